#!/usr/bin/env python3
"""Build-container check (needs /root/reference; nothing of it is copied): the REFERENCE's own test files, run with the
names `lib` and `config` bound to THIS package (`caro_ai_amd.lib`, `caro_ai_amd.config`) -- INTEGRATION level 1 ("swap
the imports") applied to the reference's test-suite.  The game-rule and helper tests need no GPU (the host-side helpers of
caro_rules.h); lib/test_mcts.py pokes a store that lives on the GPU and is skipped here (its numbers are restated in
tests/test_gpu_shim.py::TestBackup).

    python tools/ref_tests_on_this_package.py            -> pytest's summary; profiles/r05_reference_tests_on_this_package.txt
"""
import importlib
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
REF = "/root/reference"


class AliasFinder:
    """`import lib.x.y` -> `caro_ai_amd.lib.x.y`, `import config` -> `caro_ai_amd.config` (the module objects themselves)"""

    def find_spec(self, name, path=None, target=None):
        if name == "config" or name == "lib" or name.startswith("lib."):
            real = "caro_ai_amd." + name
            import importlib.util
            mod = importlib.import_module(real)
            spec = importlib.util.spec_from_loader(name, loader=self)
            spec._aliased = mod
            return spec
        return None

    def create_module(self, spec):
        return spec._aliased

    def exec_module(self, module):
        pass


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    sys.meta_path.insert(0, AliasFinder())
    import pytest
    files = [os.path.join(REF, "lib/game/connect_four/test_connect_four.py"),
             os.path.join(REF, "lib/game/tictactoe/test_tictactoe.py"),
             os.path.join(REF, "lib/game/tictactoe/test_tictactoe_helpers.py")]
    # rootdir = a scratch directory: pytest must not write into the reference tree; importmode=importlib keeps the test
    # files' own directories (which hold the reference's packages) off sys.path
    os.makedirs("/tmp/refrun", exist_ok=True)
    rc = pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", "/tmp/refrun", "--import-mode=importlib", "-c", "/dev/null"] + files)
    import lib.game.connect_four.connect_four as m
    assert m.__name__.startswith("caro_ai_amd."), m.__name__   # the tests really ran on this package's modules
    return int(rc)


if __name__ == "__main__":
    sys.exit(main())
