import ctypes, os, torch
lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lab', 'libmfma_lab.so'))
lab.lab_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
iters = 2000
for mode in range(4):
    for wg in (1, 2, 3, 4):
        for pad in (0,):
            for _ in range(2): lab.lab_run(mode, wg, iters, out.data_ptr(), pad, st)
            s, e = torch.cuda.Event(True), torch.cuda.Event(True)
            s.record(); lab.lab_run(mode, wg, iters, out.data_ptr(), pad, st); e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e)
            fl = 256.0 * wg * 4 * iters * 64 * 4096
            print('mode %d (0 regs,1 +lds reads,2 +barrier,3 +lds writes+barrier) wg/cu %d: %.3f ms %.1f TF/s' % (mode, wg, ms, fl / ms / 1e9), flush=True)
