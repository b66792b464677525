import ctypes, os, torch
lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lab', 'libmfma_bf16_lab.so'))
lab.lab_bf16.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(256, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for iters in (2000, 20000):
    for wg in (1, 2):
        for _ in range(2): lab.lab_bf16(wg, iters, out.data_ptr(), st)
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); lab.lab_bf16(wg, iters, out.data_ptr(), st); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        n = 256.0 * wg * 4 * iters * 8      # MFMAs
        print('bare bf16 32x32x16 MFMA loop, %d wg/CU, %d iters: %.3f ms  %.2f PFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)'
              % (wg, iters, ms, n * 32768 / ms / 1e12, ms * 1e-3 * 2.4e9 / (wg * iters * 8)), flush=True)

lab.lab_bf16_rand.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
rnd = torch.randn(4096 * 8, device='cuda').bfloat16().contiguous()
for iters in (2000, 20000, 100000):
    for wg in (1, 2):
        for _ in range(2): lab.lab_bf16_rand(wg, iters, out.data_ptr(), rnd.data_ptr(), st)
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); lab.lab_bf16_rand(wg, iters, out.data_ptr(), rnd.data_ptr(), st); e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        n = 256.0 * wg * 4 * iters * 8
        print('RANDOM operands, %d wg/CU, %d iters: %.3f ms  %.2f PFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)'
              % (wg, iters, ms, n * 32768 / ms / 1e12, ms * 1e-3 * 2.4e9 / (wg * iters * 8)), flush=True)
