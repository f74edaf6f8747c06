"""bf16 RN engine against the fp32 RN engine at a chosen image size / batch (conditioning of BatchNorm statistics)."""
import dataclasses, sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine_rn import create_engine

size, bs = int(sys.argv[1]), int(sys.argv[2])
mcfg = C.rn_tiny(rank=4, num_groups=2)
mcfg = dataclasses.replace(mcfg, vision=dataclasses.replace(mcfg.vision, image_size=size))
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
batch = synth.make_batch(mcfg, bs, seed=1234)
args = (batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
e32 = create_engine(mcfg, sd, dtype=torch.float32, max_images=bs)
e16 = create_engine(mcfg, sd, dtype=torch.bfloat16, max_images=bs)
o32, o16 = e32.forward_backward(*args), e16.forward_backward(*args)
print("loss", float(o32["loss"]), float(o16["loss"]))
print("logits err", float((o32["logits"] - o16["logits"]).abs().max() / o32["logits"].abs().max()))
def cmp(name, a, b):
    a, b = a.double().cpu(), b.double().cpu()
    print(f"{name:28s} max-rel {float((a-b).abs().max()/a.abs().max()):.2e}  rms-rel {float((a-b).norm()/a.norm()):.2e}")
r1 = bs * e32.H1 ** 2
for i in range(3):
    cmp(f"stem a{i+1}", e32.sa[i][:r1], e16.sa[i][:r1])
for i, (b32, b16) in enumerate(zip(e32.blocks, e16.blocks)):
    ro, ri = bs * b32.Hout ** 2, bs * b32.Hin ** 2
    cmp(f"block{i} out", b32.out[:ro], b16.out[:ro])
cmp("feat", e32.feat[:bs * mcfg.vision.tokens], e16.feat[:bs * mcfg.vision.tokens])
cmp("dfeat", e32.dfeat[:bs * mcfg.vision.tokens], e16.dfeat[:bs * mcfg.vision.tokens])
for i in range(len(e32.blocks) - 1, -1, -1):
    b32, b16 = e32.blocks[i], e16.blocks[i]
    cmp(f"block{i} dx", b32.dx[:bs * b32.Hin ** 2], b16.dx[:bs * b32.Hin ** 2])
worst = 1.0
for k in e32.params.keys:
    a, b = e32.params.view(k, "grad").double().flatten(), e16.params.view(k, "grad").double().flatten()
    c = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
    worst = min(worst, c)
print("worst grad cosine", worst)

# ---- backward alone: give the bf16 engine the fp32 engine's saved activations (rounded) and dfeat
def cp(dst, src):
    if dst is not None:
        dst.copy_(src.to(dst.dtype))
for i in range(3):
    cp(e16.sz[i], e32.sz[i]); cp(e16.sa[i], e32.sa[i])
cp(e16.p0, e32.p0)
for bn16, bn32 in zip(e16.bns, e32.bns):
    cp(bn16.mean, bn32.mean); cp(bn16.rstd, bn32.rstd)
for b16, b32 in zip(e16.blocks, e32.blocks):
    for n in ("z1", "a1", "z2", "a2", "a2p", "z3", "out", "zd"):
        if getattr(b32, n, None) is not None:
            cp(getattr(b16, n), getattr(b32, n))
    for s16, s32 in ((b16.c1, b32.c1), (b16.c3, b32.c3)):
        cp(s16.t, s32.t); cp(s16.ts, s32.ts)
for n in "qkvc":
    cp(e16.ap[n].t, e32.ap[n].t); cp(e16.ap[n].ts, e32.ap[n].ts)
for n in ("tok", "qkv", "att_o", "lse", "dfeat"):
    cp(getattr(e16, n), getattr(e32, n))
with torch.no_grad():
    e16._vision_backward(bs, 1, True)
print("--- backward only, identical saved activations")
rows = bs * mcfg.vision.tokens
cmp("d_o", e32.d_o[:rows], e16.d_o[:rows])
cmp("dqkv", e32.dqkv[:rows], e16.dqkv[:rows])
cmp("dx4", e32.dx4[:bs * mcfg.vision.spacial ** 2], e16.dx4[:bs * mcfg.vision.spacial ** 2])
for i in range(len(e32.blocks) - 1, -1, -1):
    b32, b16 = e32.blocks[i], e16.blocks[i]
    ri, ro = bs * b32.Hin ** 2, bs * b32.Hout ** 2
    cmp(f"block{i} dz3", b32.dz3[:ro], b16.dz3[:ro])
    cmp(f"block{i} da2p", b32.da2p[:ro], b16.da2p[:ro])
    cmp(f"block{i} dz2", b32.dz2[:ri], b16.dz2[:ri])
    cmp(f"block{i} da1", b32.da1[:ri], b16.da1[:ri])
    cmp(f"block{i} dz1", b32.dz1[:ri], b16.dz1[:ri])
    cmp(f"block{i} dx", b32.dx[:ri], b16.dx[:ri])
worst = 1.0
for k in e32.params.keys:
    if k.startswith("prompt"):
        continue
    a, b = e32.params.view(k, "grad").double().flatten(), e16.params.view(k, "grad").double().flatten()
    c = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
    if c < 0.999:
        print(f"  cos {c:.4f} {k}")
    worst = min(worst, c)
print("worst grad cosine", worst)
