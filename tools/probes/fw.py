import sys, importlib
sys.path[:0]=['/root/repo','/root/repo/tools']
import torch, corpus
hip = importlib.import_module("nim-snappy_amd")
nb=4096
dev=torch.device("cuda",0)
ctx=hip.Context(0)
d_in = corpus.make_blocks_torch(torch, 0, nb, dev).reshape(-1)
cap = hip.max_compressed_len_framed(nb*65536)
d_s = torch.empty(cap, dtype=torch.uint8, device=dev)
fl = ctx.compress_framed(d_in, nb*65536, d_s, cap)
d_o = torch.empty(nb*65536, dtype=torch.uint8, device=dev)
print(ctx.uncompress_framed(d_s, fl, d_o, nb*65536), fl)
