#!/usr/bin/env python3
"""Runs ON THE GPU BOX: take a failing configuration of tests/fuzz_parity.py apart.  Re-runs configuration `ci` of seed `seed`; when it
fails, rebuilds engine and oracle from the state in front of the failing rollout / step and replays its transitions ONE AT A TIME,
comparing outputs and the whole state after each — prints the first env that diverges with its record before and after on both sides,
the action, and the board around the piece.  (Round 6: this is how the O-piece-below-the-floor + hold-swap corner was read off.)
usage: python tools/fuzz_bisect.py <seed> <ci>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from tests import fuzz_parity as F  # noqa: E402
from tetris_gymnasium_amd import Engine, _native as N  # noqa: E402

seed, ci = int(sys.argv[1]), int(sys.argv[2])
keep = {}
for name in ("rollout", "step"):
    orig = getattr(Engine, name)

    def wrapped(self, acts, *a, _o=orig, _n=name, **k):
        keep["pre"], keep["acts"], keep["kind"] = self.get_state(), np.array(acts, copy=True).reshape(-1, self.B), _n
        return _o(self, acts, *a, **k)

    setattr(Engine, name, wrapped)
einit, oinit = Engine.__init__, po.OracleBatch.__init__


def einit_w(self, *a, **k):
    keep["ekw"] = (a, k)
    return einit(self, *a, **k)


def oinit_w(self, *a, **k):
    keep["okw"] = (a, k)
    return oinit(self, *a, **k)


Engine.__init__, po.OracleBatch.__init__ = einit_w, oinit_w
try:
    print(F.run_config(seed, ci, F.draw_rtc_sizes(seed)))
except AssertionError as ex:
    print("FAIL:", ex)
    (ea, ek), (oa, ok_) = keep["ekw"], keep["okw"]
    os.environ["TG_PACKED"] = "1" if " packed" in str(ex) else "0"
    e2, o2 = Engine(*ea, **ek), po.OracleBatch(*oa, **ok_)
    b, r, g = keep["pre"]
    s0 = np.arange(1, e2.B + 1, dtype=np.uint64)
    e2.reset(s0); o2.reset(s0)
    e2.set_state(b, r, g); o2.set_state(b, r, g)
    ar = {N.AUTORESET_NONE: po.AUTORESET_NONE, N.AUTORESET_SAME_STEP: po.AUTORESET_SAME_STEP, N.AUTORESET_NEXT_STEP: po.AUTORESET_NEXT_STEP}[e2.autoreset]
    print("replay from the state in front of the failing", keep["kind"], "of", len(keep["acts"]), "transitions; kernel", e2.step_kernel_name())
    for t, a in enumerate(keep["acts"]):
        pb, pr, _ = e2.get_state()
        ge = e2.step(a.astype(np.int32))
        go = o2.grouped_step(a.astype(np.int32), autoreset=ar)[:3] if ek.get("grouped") else o2.step(a.astype(np.int32), autoreset=ar)
        be, re_, _ = e2.get_state(); bo, ro, _ = o2.get_state()
        bad = [i for i in range(e2.B) if (be[i] != bo[i]).any() or ge[0][i] != go[0][i] or ge[1][i] != go[1][i] or ge[2][i] != go[2][i]
               or any(re_[f][i].tolist() != ro[f][i].tolist() for f in re_.dtype.names if f != "reserved")]
        if bad:
            print("first divergence at single step", t, "envs", bad[:8], "of", len(bad))
            for i in bad[:3]:
                print(" env", i, "action", int(a[i]), "\n   rec before ", pr[i], "\n   engine after", re_[i], "out", ge[0][i], ge[1][i], ge[2][i],
                      "\n   oracle after", ro[i], "out", go[0][i], go[1][i], go[2][i])
                x = int(pr["x"][i]); lo, hi = max(0, x - 5), min(e2.W, x + 1)
                print("   board before, columns", lo, "..", hi - 1, "rows 0..7:\n", pb[i][:8, lo:hi])
                print("   board cells that differ after the step:", np.argwhere(be[i] != bo[i])[:8].tolist())
            break
    else:
        print("single steps agree: the divergence needs the fused launch (or the operation in front of it)")
