"""Qwen2.5-Omni-7B input towers at their true shapes (random-init weights): vision tower on one image, audio tower on one clip.
python scripts/bench_towers.py [side_px=448] [audio_seconds=30]"""
import sys, torch
from spider_amd.qwen_omni import AudioTowerConfig, AudioTowerEngine, VisionTowerConfig, VisionTowerEngine
dev = torch.device("cuda:0")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 448
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
g = torch.Generator(device=dev).manual_seed(0)


def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


vc = VisionTowerConfig.qwen25_omni_7b()
vis = VisionTowerEngine.random_init(vc, dev, seed=1)
gh = side // vc.patch // 2 * 2
grid = [[1, gh, gh]]
T = gh * gh
px = torch.randn(T, vc.patch_dim, generator=g, device=dev)
ms = timed(lambda: vis(px, grid))
H, I = vc.hidden, vc.inter
fl = T * (2 * vc.patch_dim * H) + vc.depth * T * (2 * H * 3 * H + 2 * H * H + 2 * H * 2 * I + 2 * I * H)
win = 64
fl_attn = sum(4 * T * (T if l in vc.fullatt else win) * H for l in range(vc.depth))
fl += fl_attn + (T // 4) * (2 * (4 * H) ** 2 + 2 * 4 * H * vc.out_hidden)
print(f"vision tower {side}px: {T} patches -> {T // 4} tokens, {ms:.2f} ms, {fl / ms / 1e9:.0f} TF/s ({fl / 1e9:.1f} GF)")

ac = AudioTowerConfig.qwen25_omni_7b()
aud = AudioTowerEngine.random_init(ac, dev, seed=2)
frames = int(secs * 100)
feats = torch.randn(ac.mel, frames, generator=g, device=dev)
ms = timed(lambda: aud(feats, [frames]))
Ta = (frames - 1) // 2 + 1
D = ac.d_model
fl = 2 * frames * 3 * ac.mel * D + 2 * Ta * 3 * D * D + ac.layers * Ta * (2 * D * 3 * D + 2 * D * D + 4 * D * ac.ffn + 4 * 100 * D) \
    + (Ta // 2) * 2 * D * ac.out_dim
print(f"audio tower {secs:.0f}s: {frames} mel frames -> {Ta // 2} tokens, {ms:.2f} ms, {fl / ms / 1e9:.0f} TF/s ({fl / 1e9:.1f} GF)")
