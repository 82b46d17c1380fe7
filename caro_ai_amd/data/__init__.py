"""Data files of the package: `weights/` holds the checkpoints the reference ships (`saves/*/best_*.dat` =
torch.save(state_dict), lib/model.py + train.py:214-216) -- the nets bench.py, play.py's examples and the parity
tests run.  tests/golden/weights is a link to the same directory."""
import os

WEIGHTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights")


def weights_path(name):
    return os.path.join(WEIGHTS, name)
