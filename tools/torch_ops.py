"""Which torch (non-library) GPU ops does one C3 forward issue?  (glue kernels cost a launch gap each)"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtracker_amd import synth
from mvtracker_amd.tracker import MVTracker
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model = MVTracker(hidden_size=256).eval()
sd = synth.make_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev); model.precision = "bf16"
clip = synth.make_clip(1234, V=4, T=24, H=512, W=512, N=1024)
a = [torch.from_numpy(clip[k]).to(dev) for k in ("rgbs", "depths", "query_points", "intrs", "extrs")]
model(*a, iters=4); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    model(*a, iters=4); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::stack", "aten::index", "aten::index_put_", "aten::sigmoid", "aten::mul", "aten::add", "aten::ge", "aten::repeat", "aten::expand", "aten::zeros", "aten::full", "aten::arange"):
        st = [s for s in (e.stack or []) if "mvtracker_amd" in s]
        cnt[(e.name, st[0].split("/")[-1] if st else "?")] += 1
for k, v in cnt.most_common(40):
    print(v, k)
