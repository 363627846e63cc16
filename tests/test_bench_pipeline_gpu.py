"""GPU: the bench's pipelined step (encode of one batch on a side stream while p_sample advances the other batch on the main
stream, joined per step) must give, bit for bit, what the same work gives on ONE stream -- every cross-stream hazard
(cached operand images built on the other stream, workspaces, the allocator handing a block to the other stream too early)
would show here.  The reference has no pipeline (diff.py:85-127 runs the two phases back to back)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_pipelined_steps_equal_one_stream_steps():
    import bench
    w = bench.build_workload(512, 1000, 1432, "cuda:0")
    steps = 12
    with torch.no_grad():
        # one stream: step i encodes batch i % 2 and advances batch (i + 1) % 2, as the pipelined step does
        ref = []
        for i in range(steps):
            be, bd = w["batches"][i % 2], w["batches"][(i + 1) % 2]
            torch.manual_seed(100 + i)
            h = w["enc"].encode(be["x"], be["adj"])
            pred, x0 = w["diff"].p_sample(bd["img"], None, 1, bd["direction"], bd["labels"], bd["u0"], 900 - i, None, True)
            ref.append((h.clone(), pred.clone()))
        torch.cuda.synchronize()
        w["side_stream"] = torch.cuda.Stream()
        got = []
        for i in range(steps):                                  # issued without any synchronisation in between
            torch.manual_seed(100 + i)
            h, pred = bench.one_step(w, 900 - i, None, i)
            # copies on the main stream (behind the join), then h is DROPPED as the bench drops it: its block goes back to
            # the side stream's pool with no record_stream, and the next step's encode may take it -- behind that step's
            # side.wait_stream(main), i.e. behind this copy
            got.append((h.clone(), pred.clone()))
            del h, pred
        torch.cuda.synchronize()
    for i, ((h0, p0), (h1, p1)) in enumerate(zip(ref, got)):
        assert torch.equal(h0, h1), ("encode", i)
        assert torch.equal(p0, p1), ("p_sample", i)
