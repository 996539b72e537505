import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch
import whitebox_amd as W
from whitebox_amd import synth
K = 256
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    SB = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
    WL = sys.argv[2] if len(sys.argv) > 2 else "c3"
    eng, seed, amp = b.build_device_session(W, synth, WL, 4096, K, SB, 0, stream.cuda_stream, 0)
    host = torch.zeros(K * 2 * 512, dtype=torch.float32).pin_memory()
    eng.ctx.set_master_target(host.data_ptr())
    eng.play()
    done = 0
    for chunk in range(30):
        eng.ctx.kernel_time(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rew = 0
        for s in range(4):
            if done + K > SB:
                eng.stop(); eng.play(); done = 0; rew += 1
            eng.render(K); done += K
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ms, n = eng.ctx.kernel_time()
        print("chunk %2d steps %3d..%3d  step %.3f ms  mix %.3f ms  rewinds %d  done %d" % (chunk, chunk*4, chunk*4+3, dt/4*1e3, ms, rew, done))
