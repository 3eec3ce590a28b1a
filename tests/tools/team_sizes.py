"""The headline mesh (BASELINE configs[1]: torus 708 x 708, -l1 -q14) through the chunked profile with the reconstruction chain's
team size of this process (HRY_CHAIN_WAVES, read once per process) -- pipelined decode and, with HRY_NO_PIPELINE=1, one launch over
the whole chain -- against the decode of the reference-format stream of the same mesh.  At this size every kind of tile occurs
(tiles without heads, one head between two runs, heads from candidate rows, tiles prepared late with one to W - 1 tiles of overlap,
dense tiles), which the random small meshes of chain_stress.py do not guarantee.   python tests/tools/team_sizes.py [side]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from harry_amd import codec as hc
from harry_amd import meshgen as mg

side = int(sys.argv[1]) if len(sys.argv) > 1 else 708
g = mg.torus(side, side, seed=2, sigma=1e-4)
cx = hc.Codec(0)
m = hc.Mesh.from_arrays(g.verts, g.degrees, g.indices)
cx.requant(m, [(1, -1, 14)])
chunked = cx.write_hry(m.clone(), profile=hc.PROFILE_CHUNKED)
compat = cx.write_hry(m.clone(), profile=hc.PROFILE_COMPAT)
ref = cx.read_hry(compat)
for it in range(3):   # (the hand-overs between wavefronts are a matter of timing: more than one pass)
    d = cx.read_hry(chunked)
    assert (d.nv, d.nf, d.ne) == (ref.nv, ref.nf, ref.ne)
    assert np.array_equal(d.org(), ref.org())
    for l in (0, 1):
        assert np.array_equal(d.list_data(l), ref.list_data(l)), l
print("all equal")
