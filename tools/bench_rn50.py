"""RN50 backbone (BASELINE.json configs[4]: RN50 FairLoRA r=8, gender = 2 groups): step time at full size.
usage: python3 tools/bench_rn50.py [bs] [steps] [dtype] [--check] [--json]"""
import copy, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
JSON = "--json" in sys.argv
sys.argv = [a for a in sys.argv if a != "--json"]
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine_rn import create_engine

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.bfloat16
mcfg = C.rn50(rank=8, num_groups=2)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
batch = synth.make_batch(mcfg, bs, seed=1234)
args = (batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
t0 = time.time()
eng = create_engine(mcfg, sd, dtype=dtype, max_images=bs)
torch.cuda.synchronize()
if os.environ.get("FFM_SERIAL"):                       # one stream, one kernel at a time: clean per-kernel durations
    eng.set_overlap(False)
if not JSON:
    print(f"engine built in {time.time() - t0:.1f}s, {torch.cuda.memory_allocated() / 2**30:.2f} GiB, "
          f"{eng.params.numel} trainable elements")
if "--check" in sys.argv:
    from oracle import fairlora_oracle as O
    keys = synth.trainable_keys(mcfg)
    out = eng.forward_backward(*args)
    t0 = time.time()
    loss, logits, grads = O.loss_and_grads(copy.deepcopy(sd), batch, mcfg, keys)
    print(f"oracle {time.time() - t0:.1f}s  loss {float(loss):.6f} engine {float(out['loss']):.6f}")
    print("logits err", float((out["logits"].cpu() - logits).abs().max() / logits.abs().max()))
    worst = (1.0, "")
    for k in keys:
        a, b = eng.params.view(k, "grad").cpu().double().flatten(), grads[k].double().flatten()
        c = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))
        worst = min(worst, (c, k))
    print("worst gradient cosine", worst)
    sys.exit(0)
for _ in range(3):
    eng.forward_backward(*args); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    eng.forward_backward(*args); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
torch.cuda.synchronize()
ms = (time.time() - t0) / steps * 1e3
if JSON:
    print(json.dumps({"workload": f"RN50 (3,4,6,3) FairLoRA r=8 G=2, bs={bs}, 224x224x3, fwd+bwd+SGD", "ms_per_step": ms,
                      "images_per_sec": bs / ms * 1e3, "steps": steps, "dtype": str(dtype).split(".")[-1],
                      "trainable_elems": eng.params.numel, "final_loss": float(eng.loss)}))
    sys.exit(0)
print(f"RN50 r=8 G=2 bs={bs} {dtype}: {ms:.2f} ms/step, {bs / ms * 1e3:.0f} img/s, loss {float(eng.loss):.4f}")
# host enqueue time of one step (the GPU idle): how close the step is to being launch-bound
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    eng.forward_backward(*args); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
    te = time.time()
    torch.cuda.synchronize()
    t1 = time.time()
    print(f"  enqueue {1e3 * (te - t0):.2f} ms, until done {1e3 * (t1 - t0):.2f} ms")
    t0 = time.time()
