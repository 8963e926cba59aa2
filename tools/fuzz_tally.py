import os, sys, collections
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import tiktoken_amd as tts
cnt = collections.Counter()
for seed in range(150):
    rng = np.random.default_rng(21000 + seed)
    for _ in range(6):
        iw = int(rng.choice([4, 4, 2])) * int(rng.integers(8, 360)); ih = int(rng.integers(12, 420))
        ow = max(8, int(iw / rng.uniform(1.0, 2.3)))
        if rng.random() < 0.85: ow &= ~1
        oh = max(4, int(ih / rng.uniform(1.0, 6.0)))
        parts = ["scale=%d:%d%s" % (ow, oh, rng.choice(["", "", ":flags=bilinear", ":flags=area"]))]
        if rng.random() < 0.3 and ow > 40 and oh > 16:
            cw, chh = 2 * int(rng.integers(8, ow // 2 + 1)), int(rng.integers(8, oh + 1))
            parts.append("crop=%d:%d:%d:%d" % (cw, chh, 2 * int(rng.integers(0, (ow - cw) // 2 + 1)), int(rng.integers(0, oh - chh + 1))))
        fmt = str(rng.choice(["rgb24", "rgb24", "bgr24", "rgba", "bgra"]))
        n = int(rng.choice([1, 2, 5, 19]))
        try:
            plan = tts.Plan.from_vf(",".join(parts), iw, ih, in_fmt=fmt)
        except tts.TtsError as e:
            cnt["error"] += 1; continue
        cnt[plan.kernel_name(0).split("<")[0]] += 1
print(dict(cnt))
cnt = collections.Counter()
for seed in range(150):
    rng = np.random.default_rng(12000 + seed)
    for _ in range(6):
        iw, ih = 2 * int(rng.integers(40, 700)), 2 * int(rng.integers(30, 420))
        ow, oh = max(16, int(iw / rng.uniform(1.0, 2.2))), max(16, int(ih / rng.uniform(1.0, 2.2)))
        parts = ["scale=%d:%d%s" % (ow, oh, rng.choice(["", "", ":flags=bilinear", ":flags=area"]))]
        if rng.random() < 0.3:
            cw, chh = int(rng.integers(12, ow + 1)), int(rng.integers(12, oh + 1))
            parts.append("crop=%d:%d:%d:%d" % (cw, chh, int(rng.integers(0, ow - cw + 1)), int(rng.integers(0, oh - chh + 1))))
        if rng.random() < 0.3:
            parts.append("pad=iw+%d:ih+%d:-1:-1:color=%s" % (int(rng.integers(0, 40)), int(rng.integers(0, 40)), rng.choice(["black", "red", "white"])))
        parts.append("format=%s" % rng.choice(["rgb24", "rgb24", "bgr24", "rgba", "bgra"]))
        n = int(rng.choice([1, 2, 3]))
        try:
            plan = tts.Plan.from_vf(",".join(parts), iw, ih)
        except tts.TtsError:
            cnt["error"] += 1; continue
        cnt[plan.kernel_name(0).split("<")[0]] += 1
print(dict(cnt))
