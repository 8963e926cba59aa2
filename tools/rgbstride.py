import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch, tiktoken_amd as tts
from sweep import CASES, plan_for
case = sys.argv[1]; frames = int(sys.argv[2])
vf, iw, ih, algo = CASES[case]
plan = plan_for(case)
ow, oh = plan.out_size
bpp = plan.out_bytes // (ow * oh)
src = torch.randint(0, 256, (frames, plan.in_bytes), dtype=torch.uint8, device="cuda")
for stride in [ow * bpp] + [int(x) for x in sys.argv[3].split(",")]:
    dst = torch.empty((frames, oh * stride), dtype=torch.uint8, device="cuda")
    ts = []
    for rnd in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            plan.run_batch_strided(src, (iw, (iw + 1) // 2), dst, (stride, 0), frames)
        e1.record(); torch.cuda.synchronize()
        if rnd: ts.append(e0.elapsed_time(e1) / 5)
    ts.sort(); med = ts[len(ts) // 2]
    print("%s stride %d: %.4f ms  %.1f%%" % (case, stride, med, algo * frames / med / 1e6 / 80))
