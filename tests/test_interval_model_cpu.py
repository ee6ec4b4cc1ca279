"""The argument behind the packet walk's triangle masks (kernels.hip, tri_may_hit), checked on the CPU: the Moeller-Trumbore
expression sequence evaluated on intervals (corner evaluation of monotone f32 operations) never rejects a triangle that some ray
inside the bounds hits.  numpy model of tools/sim_tri_reject.py against per-ray f32 tests on camera packets of the teapot."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_interval_rejection_is_conservative():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sim_tri_reject as st
    from sim_collapse import RefTree, load

    ref = RefTree(*load("teapot", 1.0))
    rng = np.random.default_rng(5)
    packets = st.camera_packets("teapot", 8, rng, 2, w=192, h=108)
    rejected = hit = wrong = 0
    for o, d in packets:
        for pad in (0.0, 0.25):
            eo, ed = (o.max(0) - o.min(0)) * np.float32(pad), (d.max(0) - d.min(0)) * np.float32(pad)
            b = ((o.min(0) - eo).astype(np.float32), (o.max(0) + eo).astype(np.float32), (d.min(0) - ed).astype(np.float32), (d.max(0) + ed).astype(np.float32))
            for v0, e1, e2 in ref.leaf.values():
                rej = st.interval_reject(v0, e1, e2, *b)
                ok, _ = st.mt_valid(v0, e1, e2, o, d)
                some = ok.any(axis=0)
                rejected += int(rej.sum()); hit += int(some.sum()); wrong += int((rej & some).sum())
    assert wrong == 0
    assert hit > 0 and rejected > 10 * hit  # the test sees hits, and the bounds reject most of the rest
