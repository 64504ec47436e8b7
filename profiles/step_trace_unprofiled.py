#!/usr/bin/env python3
"""Every launch of ONE ADMM iteration of the shipped schedule, without a profiler (lshm_trace_*: each launch carries its
own start / stop events): start, duration, stream, totals by kernel -- the format of profiles/step_trace.py.  An iteration
here = Adam -> the two forwards side by side -> reconstruction pass -> (next call) backward, i.e. from one adam_kernel to
the next, like the rocprofv3 timelines.  Usage: python profiles/step_trace_unprofiled.py [schedule_off,...]"""
import ctypes as C, os, re, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lshm_amd import KHarmonicTrainer, TrainConfig, _lib as L

B, dev = 256, torch.device("cuda:0")
off = tuple(n for n in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if n)
cfg = TrainConfig(Kc=10, schedule_off=off, tune=int(os.environ.get("PROBE_TUNE", 0)))
tr = KHarmonicTrainer(cfg, batch=B, batch_per_bline=8, default_batch=B // 8, device=dev)
tr.init_parameters(seed=0)
gen = torch.Generator(device="cpu").manual_seed(1234)
x = torch.randn(B, cfg.num_in_channels, 128, 128, generator=gen)
x = (x - x.mean()) / x.std()
uv = 1000.0 * torch.randn(B, 2, generator=gen)
tr.new_minibatch(x.to(dev), uv.to(dev))
lib = L.load()
for _ in range(30):
    tr.step()
torch.cuda.synchronize()
# untraced reference time of the same loop
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(40):
    tr.step()
b.record(); torch.cuda.synchronize()
ref_ms = a.elapsed_time(b) / 40
ENDS_ONLY = os.environ.get("TRACE_ENDS_ONLY") == "1"
L.check(lib.lshm_trace_begin_ex(4096, 0 if ENDS_ONLY else 1), "trace_begin")
a.record()
for _ in range(6):
    tr.step()
b.record()
n = lib.lshm_trace_end()
torch.cuda.synchronize()
traced_ms = a.elapsed_time(b) / 6
recs = []
name = C.create_string_buffer(512)
st, du, sx, gt = C.c_float(), C.c_float(), C.c_int(), C.c_uint()
for i in range(n):
    L.check(lib.lshm_trace_read(i, name, 512, C.byref(st), C.byref(du), C.byref(sx), C.byref(gt)), "trace_read")
    recs.append((st.value, du.value, sx.value, gt.value, name.value.decode()))
lib.lshm_trace_free()
if ENDS_ONLY:
    # completion times only (no start packets: the queues run as untraced): per stream, completion - previous completion of
    # the stream = duration + whatever the launch waited for
    adam = [i for i, r in enumerate(recs) if "adam_kernel" in r[4]]
    it = sorted(recs[adam[2]:adam[3]], key=lambda r: r[0])
    t0 = it[0][0]
    print(f"# completion-only trace: schedule_off={off}; loop {ref_ms:.4f} ms per iteration untraced, {traced_ms:.4f} ms traced")
    print(f"# iteration (adam completion to the last completion): {max(r[0] for r in it) - t0:.1f} us, {len(it)} kernels")
    prev = {}
    for s_, d_, q, g, nm in it:
        dlt = s_ - prev.get(q, s_)
        prev[q] = s_
        print(f"{s_ - t0:9.1f} q={q:3d} since-previous-completion-on-stream={dlt:7.1f} grid={g:8d} {re.sub(r'[(]anonymous namespace[)]::', '', nm.replace('lshm::', ''))[:110]}")
    sys.exit(0)
adam = [i for i, r in enumerate(recs) if "adam_kernel" in r[4]]
lo, hi = adam[2], adam[3]  # the third traced iteration
it = sorted(recs[lo:hi], key=lambda r: r[0])
t0 = it[0][0]
end = max(r[0] + r[1] for r in it)
# union of busy intervals
busy, cur_s, cur_e = 0.0, None, None
for s_, d_, *_ in it:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s_, s_ + d_
    else:
        cur_e = max(cur_e, s_ + d_)
busy += cur_e - cur_s
print(f"# unprofiled per-launch trace: schedule_off={off}; loop time {ref_ms:.4f} ms per iteration untraced, {traced_ms:.4f} ms with the trace on")
print(f"# step wall {end - t0:.1f} us, {len(it)} kernels, device busy (union) {busy:.1f} us, sum {sum(r[1] for r in it):.1f} us")
last_end = {}
short = lambda s: re.sub(r"\(anonymous namespace\)::", "", s.replace("lshm::", ""))[:110]
for s_, d_, q, g, nm in it:
    gap = s_ - last_end.get(q, s_)
    last_end[q] = s_ + d_
    print(f"{s_ - t0:9.1f} q={q:3d} dur={d_:7.1f} gap={max(gap, 0):7.1f} grid={g:8d} {short(nm)}")
tot = {}
for s_, d_, q, g, nm in it:
    k = short(nm).split("(")[0]
    tot.setdefault(k, [0, 0.0])
    tot[k][0] += 1; tot[k][1] += d_
print("# ---- totals by kernel within the step")
for k, (c, s_) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"# {k:70s} n={c:3d} sum={s_:8.1f} us avg={s_ / c:7.1f} us")
