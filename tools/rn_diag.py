"""Per-tensor gradient error of the RN engine against the oracle (diagnostic)."""
import copy, sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine_rn import create_engine
from oracle import fairlora_oracle as O

dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float32
mcfg = C.rn_tiny(rank=4, num_groups=2)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
batch = synth.make_batch(mcfg, 6, seed=1234)
eng = create_engine(mcfg, sd, dtype=dtype, max_images=6)
out = eng.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
keys = synth.trainable_keys(mcfg)
loss, logits, grads = O.loss_and_grads(copy.deepcopy(sd), batch, mcfg, keys)
print("loss", float(out["loss"]), float(loss))
for k in keys:
    g, ref = eng.params.view(k, "grad").cpu().double(), grads[k].double()
    e = float((g - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    c = float(torch.dot(g.flatten(), ref.flatten()) / (g.norm() * ref.norm()).clamp_min(1e-300))
    print(f"{e:9.2e} {1-c:9.2e} {float(ref.abs().max()):9.2e} {k}")

# ---- intermediate tensors: block outputs and their gradients
rec = []
orig = O.bottleneck
def wrapped(*a, **k):
    y = orig(*a, **k)
    y.retain_grad()
    rec.append(y)
    return y
O.bottleneck = wrapped
loss, logits, grads = O.loss_and_grads(copy.deepcopy(sd), batch, mcfg, keys)
def nhwc(t):
    return t.detach().permute(0, 2, 3, 1).reshape(-1, t.shape[1]).double()
for i, y in enumerate(rec):
    blk = eng.blocks[i]
    rows = 6 * blk.Hout ** 2
    out = blk.out[:rows].cpu().double()
    ref = nhwc(y)
    flips = int(((out > 0) != (ref > 0)).sum())
    print(f"block {i}: out err {float((out-ref).abs().max()/ref.abs().max()):.2e}  relu flips {flips}")
    if i + 1 < len(rec):
        g = eng.blocks[i + 1].dx[:rows].cpu().double()
        gr = nhwc(y.grad)
        d = (g - gr).abs()
        print(f"   d(out) err {float(d.max()/gr.abs().max()):.2e}  at {int(d.argmax())//g.shape[1]},{int(d.argmax())%g.shape[1]} "
              f"mean err {float(d.mean()/gr.abs().mean()):.2e}")
