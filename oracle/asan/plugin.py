"""pytest plugin of the CPU sanitizer run (oracle/asan/run.sh) -- test infrastructure.

Swaps in the -fsanitize=address,undefined builds: the oracle's shared object, and -- in place of libcaro_hip.so,
whose host side hipcc compiled -- the g++ build of the product's host helpers (oracle/asan/host_tu.cpp), binding the
host subset of include/caro_hip.h only.  Tests that need anything else from the library are not selected by run.sh.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "..", "_build")
HOST_SYMBOLS = ["caro_last_error", "caro_version", "caro_key_words", "caro_action_space", "caro_obs_cells",
                "caro_host_initial", "caro_host_move", "caro_host_legal", "caro_host_encode", "caro_host_noise_row",
                "caro_host_move_uniform"]


def pytest_configure(config):
    from caro_ai_amd import _lib
    from oracle import oracle as orc
    orc._SO = os.path.join(BUILD, "libcaro_oracle_asan.so")
    orc.build = lambda force=False: orc._SO
    assert os.path.exists(orc._SO), "run `make -C oracle asan` first"
    host = C.CDLL(os.path.join(BUILD, "libcaro_host_asan.so"))
    for name in HOST_SYMBOLS:
        res, args = _lib._SIGNATURES[name]
        fn = getattr(host, name)
        fn.restype, fn.argtypes = res, args
    _lib._lib = host  # what _lib.load() hands out from now on
    print("[asan] oracle = %s, host helpers = libcaro_host_asan.so" % os.path.basename(orc._SO))
