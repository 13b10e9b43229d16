"""Debug: the no-allocation entry point eager on a side stream, then captured into a graph and replayed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import FRAMES, read_frame
from zstd_amd import api

dctx = api.DCtx(0)
entries = [e for e in FRAMES if "dict" not in e and e["csize"] > 0 and e["capacity"] <= (1 << 16)
           and not e["name"].startswith(("concat", "skippable", "rec_nodict"))][:40]
frames = [read_frame(e) for e in entries]
src, fin, dst_bytes = api.pack_frames(frames, [e["capacity"] for e in entries])
d_src = torch.from_numpy(src).cuda()
d_fin = torch.from_numpy(fin.view(np.uint8).copy()).cuda()
d_dst = torch.full((dst_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
d_st = torch.full((len(fin) * api.FRAME_STATUS_DTYPE.itemsize,), 0xEE, dtype=torch.uint8, device="cuda")

def show(tag):
    torch.cuda.synchronize()
    st = d_st.cpu().numpy().view(api.FRAME_STATUS_DTYPE)[: len(fin)]
    print(tag, [int(x) for x in st["errorCode"][:12]], flush=True)

def call(stream):
    api.decode_frames_noalloc(dctx, d_src.data_ptr(), src.nbytes, d_fin.data_ptr(), len(fin), d_dst.data_ptr(), d_st.data_ptr(), None, stream)

call(torch.cuda.current_stream().cuda_stream); show("eager default stream")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    d_st.fill_(0xEE)
    call(s.cuda_stream)
show("eager side stream")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    call(torch.cuda.current_stream().cuda_stream)
show("after capture (nothing ran)")
for i in range(2):
    d_st.fill_(0xEE)
    g.replay()
    show("replay %d" % i)
call(torch.cuda.current_stream().cuda_stream); show("eager again")
