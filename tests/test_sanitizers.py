"""CPU sanitizer leg of the suite ("-fsanitize=address host tests"): the oracle and the product's
host-side rule / noise helpers, both rebuilt with -fsanitize=address,undefined (`make -C oracle asan`), run the
rules, known-answer and table-net search tests again in a child interpreter with the sanitizer runtimes preloaded
(oracle/asan/run.sh).  The full selection -- the conv-net games too -- is `make -C oracle asan-test` (2 minutes,
result in profiles/asan_r03.txt).  The 32-game arena fixtures run their first two games each here (CARO_UNDER_ASAN).  GPU sanitizers do not exist on the pool; this runs in the CPU container only."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="needs gcc / g++")
def test_oracle_and_host_helpers_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, CARO_ASAN_K="not real_weights and not conv_net")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run(["sh", os.path.join(ROOT, "oracle", "asan", "run.sh")], env=env, capture_output=True, text=True,
                       timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail
