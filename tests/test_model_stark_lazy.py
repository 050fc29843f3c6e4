"""CPU: the Python model of the Stark lazy-limb arithmetic (tools/model_stark_lazy.py) agrees with plain integers and keeps every
column accumulator inside int64 -- the bound analysis stark_lazy.hpp relies on."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stark_lazy_model_self_check():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "model_stark_lazy.py"), "3000"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok maxacc bits" in r.stdout
    bits = int(r.stdout.split("ok maxacc bits")[1].split()[0])
    assert bits <= 63
    # the constant the header embeds for tw_from_u64 (2^560 mod p in 28-bit limbs)
    hdr = open(os.path.join(ROOT, "stark_rings_amd", "csrc", "stark_lazy.hpp")).read()
    limbs = r.stdout.split("R2 limbs (2^560 mod p):")[1].split("\n")[0]
    for tok in limbs.replace("[", "").replace("]", "").replace("'", "").split(","):
        assert tok.strip() in hdr, tok
