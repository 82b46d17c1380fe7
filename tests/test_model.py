"""Net: checkpoint format identical to the reference's; inference forms
(FoldedNet, GemmNet) equal Net.eval() within a stated float32 tolerance."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN

W_C4 = os.path.join(GOLDEN, "weights", "best_026_12000.dat")
W_TTT = os.path.join(GOLDEN, "weights", "best_005_00900.dat")


def test_state_dict_layout_matches_shipped_checkpoints():
    """62 keys, reference names and shapes (checkpoint row)."""
    from caro_ai_amd.lib.model import Net
    for path, shape, A in [(W_C4, (2, 6, 7), 7), (W_TTT, (2, 3, 3), 9)]:
        sd = torch.load(path, map_location="cpu")
        net = Net(shape, A)
        mine = net.state_dict()
        assert list(mine.keys()) == list(sd.keys()) and len(sd) == 62
        for k in sd:
            assert mine[k].shape == sd[k].shape and mine[k].dtype == sd[k].dtype, k
        net.load_state_dict(sd)  # strict


def test_dat_round_trip(tmp_path):
    """torch.save(net.state_dict(), *.dat) as train.py:214-216, load as play.py:31-33."""
    from caro_ai_amd.lib.model import Net, NetWrapper
    torch.manual_seed(1)
    net = Net((2, 6, 7), 7)
    p = tmp_path / "best_001_00100.dat"
    torch.save(net.state_dict(), str(p))
    net2 = Net((2, 6, 7), 7)
    net2.load_state_dict(torch.load(str(p), map_location=lambda storage, loc: storage))
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)
    w = NetWrapper(net)
    with torch.no_grad():
        net.policy[0].bias.add_(1.0)
    assert not torch.equal(w.target_model.policy[0].bias, net.policy[0].bias)
    w.sync()
    assert torch.equal(w.target_model.policy[0].bias, net.policy[0].bias)


def test_forward_matches_reference_recorded_outputs():
    """G4: logits / values the REFERENCE's Net produced for 256 boards (CPU fp32)."""
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    torch.set_num_threads(1)
    d = np.load(os.path.join(GOLDEN, "net_c4_forward.npz"))
    g = ConnectFour()
    net = Net(g.obs_shape, g.action_space)
    net.load_state_dict(torch.load(W_C4, map_location="cpu"))
    net.eval()
    x = torch.from_numpy(g.states_to_training_batch([int(s) for s in d["states"]], d["who"].tolist()))
    with torch.no_grad():
        lg, vl = net(x)
    # same torch build, same kernels: bit exact here; 1e-5 leaves room for another BLAS
    np.testing.assert_allclose(lg.numpy(), d["logits"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(vl.numpy(), d["values"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("form", ["FoldedNet", "GemmNet"])
@pytest.mark.parametrize("path,shape,A", [(W_C4, (2, 6, 7), 7), (W_TTT, (2, 3, 3), 9), (None, (2, 15, 15), 225)])
def test_inference_forms_match_eval_net(form, path, shape, A):
    from caro_ai_amd.lib import model
    torch.manual_seed(3)
    net = model.Net(shape, A)
    if path:
        net.load_state_dict(torch.load(path, map_location="cpu"))
    else:  # give the batch norms non-trivial statistics
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.5, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.uniform_(-0.3, 0.3)
    net.eval()
    inf = getattr(model, form)(net).eval()
    x = (torch.rand(37, *shape) < 0.3).float()
    x[:, 1] *= (1 - x[:, 0])
    with torch.no_grad():
        lg, vl = net(x)
        lg2, vl2 = inf(x)
    # float32 re-association only: stated tolerance 2e-4 absolute on logits (|logits| ~ 1..10), 2e-5 on tanh values
    assert (lg - lg2).abs().max().item() < 2e-4
    assert (vl - vl2).abs().max().item() < 2e-5
    assert lg2.shape == (37, A) and vl2.shape == (37, 1)


def test_shape_probes_of_the_reference_net():
    """lib/model.py:74-80: the feature counts the 1x1 heads hand to their linear layers"""
    from caro_ai_amd.lib.model import Net
    for shape, A in (((2, 6, 7), 7), ((2, 3, 3), 9), ((2, 15, 15), 225)):
        net = Net(shape, A)
        body = (64, shape[1], shape[2])
        assert net._get_conv_val_size(body) == shape[1] * shape[2] == net.value[0].in_features
        assert net._get_conv_policy_size(body) == 2 * shape[1] * shape[2] == net.policy[0].in_features
