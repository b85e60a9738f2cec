"""Host-fed loop for 1000 steps: pace and device memory (Prefetcher.max_ahead = $MAX_AHEAD, 0 = unbounded)."""
import importlib, os, sys, time, torch
sys.path.insert(0, "/root/repo")
gs = importlib.import_module("3dvlp_amd.grounding_step"); synth = importlib.import_module("3dvlp_amd.synth"); ip = importlib.import_module("3dvlp_amd.input_pipeline")
dev = torch.device("cuda:0")
cs = torch.cuda.Stream()
step = gs.GroundingStep(dev, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True)
with torch.cuda.stream(step._side): torch.zeros(1, device=dev)
with torch.cuda.stream(cs): torch.zeros(1, device=dev)
host = []
for j in range(3):
    hb = ip.compress_cloud({k: torch.from_numpy(v) for k, v in synth.make_batch(8 * j, 8, 40000, 8).items()})
    host.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in hb.items()})
def endless():
    i = 0
    while True:
        yield host[i % 3]; i += 1
feed = ip.Prefetcher(endless(), device=dev, prepare=gs.prepare_batch, stream=cs, max_ahead=int(os.environ.get("MAX_AHEAD", 3)))
cur, nxt = feed.next(), feed.next()
t0 = time.perf_counter()
for i in range(1, 1001):
    step.run(cur, nxt); cur, nxt = nxt, feed.next()
    if i % 250 == 0:
        th = time.perf_counter() - t0
        print(f"step {i}: host {1e3*th/i:.2f} ms/step so far, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
torch.cuda.synchronize()
print(f"done {1e3*(time.perf_counter()-t0)/1000:.3f} ms/step, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB")
