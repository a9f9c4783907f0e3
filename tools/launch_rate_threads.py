"""System-wide kernel launch rate: n host threads (one stream each) vs n processes, empty 256-block kernels and the decode-like
touch kernel. If threads do not scale and processes do, the in-flight decode is bound by a lock in the HIP runtime."""
import os, sys, time, threading, subprocess, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lip2speech_amd import native
L = native.lib()
N = 20000

def chain(stream, kind, buf, out, n=N):
    # kind 0: empty kernel; kind 1: 128 blocks each streaming 16 x 8 KiB (a decode-phase stand-in, ~2 us)
    native.check(L.l2s_op_launch_chain(kind, n, 256 if kind == 0 else 128, 0 if kind == 0 else 16, buf, out, stream))

def one_process(nthreads, kind, tag=""):
    import ctypes
    xs = [torch.zeros(48 << 18, device="cuda") for _ in range(nthreads)]
    ys = [torch.zeros(1 << 16, device="cuda") for _ in range(nthreads)]
    ss = [torch.cuda.Stream() for _ in range(nthreads)]
    fp = ctypes.POINTER(ctypes.c_float)
    def run(i, n):
        native.check(L.l2s_op_launch_chain(kind, n, 256 if kind == 0 else 128, 0 if kind == 0 else 16,
                                           ctypes.cast(xs[i].data_ptr(), fp), ctypes.cast(ys[i].data_ptr(), fp), ctypes.c_void_p(ss[i].cuda_stream)))
    for i in range(nthreads): run(i, 200)
    torch.cuda.synchronize()
    th = [threading.Thread(target=run, args=(i, N)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{tag}kind {kind}: {nthreads} thread(s): host {t_host / N * 1e6:6.2f} us per launch-per-thread, wall {t / N * 1e6:6.2f}; "
          f"system-wide {nthreads * N / t / 1e3:7.1f} k launches/s = {t / (nthreads * N) * 1e6:5.2f} us/launch", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        time.sleep(max(0.0, float(sys.argv[3]) - time.time()))      # common start time
        one_process(1, int(sys.argv[1]), tag=f"[proc {sys.argv[2]}] ")
        sys.exit(0)
    for kind in (0, 1):
        for nt in (1, 2, 3, 4):
            one_process(nt, kind)
    for kind in (0, 1):
        for npr in (2, 3):
            print(f"--- {npr} processes, one thread each")
            start = time.time() + 25
            ps = [subprocess.Popen([sys.executable, __file__, str(kind), str(i), str(start)]) for i in range(npr)]
            for p in ps: p.wait()
