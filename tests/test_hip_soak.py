"""GPU: soak.  Two contexts are fed the same 1500-frame sequence through the asynchronous entry point (float64 host clouds of two sizes,
so both scatter paths and both ray-kernel variants run; visibility pass on; the map moves every few frames; variance / time updates in
between) and must stay bit-identical at every checkpoint: nothing on the frame path -- ticket counters, pipelined uploads, queued ray work,
pending map moves -- may depend on timing."""
import hashlib

import numpy as np
import pytest

import _fixtures as fx
from _util import make_pair
from oracle import emap_oracle as eo

pytestmark = pytest.mark.gpu


def test_two_contexts_stay_bit_identical_over_many_frames(weights):
    C, FRAMES = 512, 1500
    rng = np.random.default_rng(5)
    big = [fx.cloud(C, 300_000, s, dz=-0.01 * s).astype(np.float64) for s in range(4)]
    small = [fx.cloud(C, 40_000, 10 + s, dz=-0.01 * s).astype(np.float64) for s in range(4)]
    moves = rng.integers(-4, 5, size=(FRAMES, 2))
    R, t = fx.POSES["rotated"]
    digests = []
    for run in range(2):
        hip, _ = make_pair(eo.YAML, C, "reference_fp16", weights)
        buf_b, buf_s = np.empty_like(big[0]), np.empty_like(small[0])
        seq = []
        for f in range(FRAMES):
            src, buf = (big, buf_b) if f % 3 else (small, buf_s)
            buf[...] = src[f % 4]
            hip.input_pointcloud(buf, ["x", "y", "z"], R, t.copy() + hip.center, 1.0, 1.0)
            buf[...] = np.nan                                      # the caller's buffer is free again
            if f % 5 == 0:
                hip.shift_map_xy(moves[f])
            if f % 7 == 0:
                hip.update_variance(); hip.update_time()
            if f % 250 == 249:
                seq.append(hashlib.sha1(hip.elevation_map.tobytes() + hip.normal_map.tobytes()).hexdigest())
        digests.append(seq)
        hip.close()
    assert digests[0] == digests[1] and len(set(digests[0])) == len(digests[0])
