"""The data-parallel step's stream / event / collective structure on the one GPU a box has: backend nccl (RCCL) with
a one-rank group, TrainStep told world_size = 2 (bucket events, communication stream, reduce-scatter, sharded Adam,
all-gather -- the collectives move nothing, their launches and orderings are real).  TOLD_WORLD=1: the group is alive
and the step takes the single-GPU path (what a communicator costs by existing).  Times the step under the
CRADLE_* settings given on the command line: python tools/bench_ddp_one_rank.py [steps]"""
import os, sys, time
if "CRADLE_HW_QUEUES" in os.environ:  # what bench.py does for N > 1 (default there: 16); unset here = the runtime's 4
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["CRADLE_HW_QUEUES"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep

shape = syn.SHAPES[os.environ.get("SHAPE", "norman")]
lm, batch = bench.build(shape, "cuda:0", seed=1000)
# EXPERIMENT=padK: K streams created and used once before TrainStep makes its communication stream (moves that stream
# to another hardware queue)
_pad = next((int(e[3:]) for e in os.environ.get("EXPERIMENT", "").split(",") if e.startswith("pad")), 0)
_pads = [torch.cuda.Stream() for _ in range(_pad)]
for _s in _pads:
    with torch.cuda.stream(_s):
        torch.zeros(1, device="cuda:0")
torch.cuda.synchronize()
st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0, world_size=int(os.environ.get("TOLD_WORLD", "2")))
if os.environ.get("MAIN_STREAM") == "pool":  # the step on a stream of torch's pool instead of the device's null stream
    main = torch.cuda.Stream()
    main.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main)
# EXPERIMENT (comma-separated), to take the two-bucket form apart without a profiler (which slows the host and, with it,
# hides the stall): nowait = the communication stream does not wait for the step's bucket event (results wrong, timing
# only); nocoll = no reduce-scatter / all-gather calls at all (the wait stays); waitonly = the early bucket's
# collective replaced by nothing, its wait and the hand-back to the step's stream kept; syncK = the host waits for the
# device every K steps (bounds how far it runs ahead)
exp = [e for e in os.environ.get("EXPERIMENT", "").split(",") if e]
sync_every = next((int(e[4:]) for e in exp if e.startswith("sync")), 0)
if "nowait" in exp:
    from cradle_vae_amd import hip as _hip
    _hip.load().cradle_stream_wait_event = lambda *a: 0
if "nocoll" in exp:
    from cradle_vae_amd import parallel as _par
    _par.reduce_scatter_inplace_ = lambda g, rk, gw: None
    _par.all_gather_inplace_many_ = lambda ts, rk, gw: None
if "waitonly" in exp:
    from cradle_vae_amd import parallel as _par
    _orig_rs = _par.reduce_scatter_inplace_
    _par.reduce_scatter_inplace_ = lambda g, rk, gw: None if torch.cuda.current_stream() == st.comm_stream else _orig_rs(g, rk, gw)
for _ in range(20):
    st(batch, n_particles=shape.n_particles)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    st(batch, n_particles=shape.n_particles)
    if sync_every and i % sync_every == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
host = []
for _ in range(10):
    t1 = time.perf_counter()
    for _ in range(8):
        st(batch, n_particles=shape.n_particles)
    host.append((time.perf_counter() - t1) / 8 * 1e3)
    torch.cuda.synchronize()
host.sort()
print(f"host enqueue {host[len(host)//2]:.3f} ms/step (bursts of 8 after a sync)")
print(f"{shape.name} experiment={','.join(exp) or '-'} buckets={os.environ.get('CRADLE_BUCKETS', '1')} sync={st.grad_sync} overlap={st.overlap} async_allgather={getattr(st, 'async_allgather', None)}: "
      f"{dt*1e3:.4f} ms/step = {shape.n_cells/dt:.0f} cells/s per rank", flush=True)
dist.destroy_process_group()
