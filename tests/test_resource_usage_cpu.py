"""The compiler's per-kernel resource report of the built library (ntt_amd/csrc/Makefile leaves it beside
every policy object): the kernels large transforms run on -- the 8-stage passes (k1 = k2 = 4), the
fused tails and the fused product launch -- use no scratch memory for any arithmetic policy (a
register spill in a butterfly kernel costs more than any tuning gains) and keep their four workgroups
(16 waves) per CU, the product launch its three.  The generic rounds with 5 to 7 stages (n = 2^5..2^7,
the first pass of n = 2^14 / 2^22 and of odd sizes) hold a radix-16 round's twiddles in registers and
may spill a few words; that is bounded here so it cannot grow unnoticed."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORTS = sorted(glob.glob(os.path.join(ROOT, "build", "obj", "policy_*.usage.txt")))


def kernels(path):
    out, cur = [], None
    for line in open(path):
        m = re.search(r"remark: (?:[^\s]+: )?(.+?) \[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            cur = {"name": text.split(":", 1)[1].strip()}
            out.append(cur)
        elif cur is not None and ":" in text:
            k, v = text.split(":", 1)
            cur[k.strip()] = v.strip()
    return out


@pytest.mark.skipif(not REPORTS, reason="no resource reports (build/ is not shipped to GPU boxes; run make first)")
def test_no_kernel_spills_and_occupancy_holds():
    assert len(REPORTS) == 6, REPORTS
    for path in REPORTS:
        ks = kernels(path)
        assert len(ks) >= 30, (path, len(ks))
        for k in ks:
            name = k["name"]
            scratch = int(k["ScratchSize [bytes/lane]"])
            generic = re.search(r"Li4ELi[123]ELb", name) is not None  # pass_kernel<..., 4, 1|2|3, ...>
            assert scratch <= (64 if generic else 0), (os.path.basename(path), name, k)
            occ = int(k["Occupancy [waves/SIMD]"])
            if "product_mid_kernel" in name:
                assert occ >= 3, (name, k)
            elif "pass_kernel" in name or "fused_small_kernel" in name:
                assert occ >= 4, (name, k)
