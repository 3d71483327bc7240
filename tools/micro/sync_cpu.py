#!/usr/bin/env python3
"""How much host CPU does waiting for the GPU cost?  (BENCH_r02: 35 ms of host CPU per 114-ms step with 20 steps queued.)
Process CPU time (all threads) while ~1 s of GPU work drains, for: torch.cuda.synchronize(), a blocking-sync event, a sleep-poll on
event.query(), and the first two again after hipSetDeviceFlags(hipDeviceScheduleBlockingSync)."""
import ctypes, time, torch
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
def work(n=60):
    for _ in range(n): torch.mm(x, x)
def measure(name, wait):
    work(5); torch.cuda.synchronize()
    work()
    c0, t0 = time.process_time(), time.perf_counter()
    wait()
    c1, t1 = time.process_time(), time.perf_counter()
    print(f"{name:50s} wall {1e3 * (t1 - t0):7.1f} ms   process CPU {1e3 * (c1 - c0):7.1f} ms", flush=True)
def ev_block():
    e = torch.cuda.Event(blocking=True); e.record(); e.synchronize()
def ev_poll():
    e = torch.cuda.Event(); e.record()
    while not e.query(): time.sleep(0.0005)
measure("torch.cuda.synchronize()", torch.cuda.synchronize)
measure("Event(blocking=True).synchronize()", ev_block)
measure("sleep-poll on Event.query() (0.5 ms)", ev_poll)
hip = ctypes.CDLL("libamdhip64.so")
print("hipSetDeviceFlags(BlockingSync) ->", hip.hipSetDeviceFlags(4))
measure("torch.cuda.synchronize() after the flag", torch.cuda.synchronize)
measure("Event(blocking=True).synchronize() after the flag", ev_block)
