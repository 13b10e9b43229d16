#!/usr/bin/env python3
"""Stage times of one bench configuration without the checks of bench.py -- for experimental builds of the library
(ZSTDGPU_LIB=<file>) whose output is allowed to be wrong (a kernel with a part switched off to see what it costs).

    python tools/time_stages.py [--config 2] [--frames N] [--steps 10]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2")
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    import bench
    from zstd_amd import api
    w = bench.make_workload(args.config, 0, 1, args.frames, 0, 8.0)
    dctx = api.DCtx(0)
    ddict = api.DDict(dctx, w["dictionary"]) if w["dictionary"] else None
    src, fin, dst_bytes = api.pack_frames(w["frames"], w["caps"])
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.zeros(dst_bytes, dtype=torch.uint8, device="cuda")
    b = api.Batch(dctx, src, fin, ddict)
    b.set_profiling(True)
    stream = torch.cuda.current_stream().cuda_stream
    acc = np.zeros(api.NUM_STAGES)
    for i in range(args.steps + 2):
        b.decode(d_src.data_ptr(), d_dst.data_ptr(), stream)
        torch.cuda.synchronize()
        if i >= 2:
            acc += np.array(b.stage_ms())
    acc /= args.steps
    print(" ".join(f"{api.STAGE_NAMES[i]} {acc[i]:.4f}" for i in range(api.NUM_STAGES)))
    st = b.status()
    print("frames with an error:", int((st["errorCode"] != 0).sum()), "of", len(st))


if __name__ == "__main__":
    main()
