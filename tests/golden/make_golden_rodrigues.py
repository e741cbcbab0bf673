#!/usr/bin/env python3
"""Generate tests/golden/rodrigues_twin.npz with the reference's OWN in-repo axis-angle -> matrix code
(common/rot.py:316-327 batch_rodrigues + quat_to_rotmat; the same function is repeated in
src/models/handoccnet_light/mano_head.py:5-16).  smplx's batch_rodrigues (the one a9 uses) is a third-party
dependency that is absent; this in-repo twin is the same map written through quaternions, so the two must
agree to fp32 rounding.  Dev container only."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_shims import *  # noqa: F401,F403
from _ref_shims import META, ref_rot
import numpy as np
import torch


def main():
    g = torch.Generator().manual_seed(77)
    rv = torch.randn(512, 3, generator=g)
    rv[:64] *= 0.01                                   # small angles
    rv[64:128] *= 3.0                                 # beyond pi
    rv[128] = 0.0                                     # the +1e-8 guard
    rv[129] = torch.tensor([3.14159, 0.0, 0.0])
    out32 = ref_rot.batch_rodrigues(rv)
    out64 = ref_rot.batch_rodrigues(rv.double())
    meta = dict(META, what="common/rot.py batch_rodrigues (quaternion form) on 512 rotation vectors, fp32 and fp64")
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rodrigues_twin.npz"),
                        rotvec=rv.numpy(), R32=out32.numpy(), R64=out64.numpy(), meta=np.array(json.dumps(meta)))
    print("wrote rodrigues_twin.npz")


if __name__ == "__main__":
    main()
