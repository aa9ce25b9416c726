"""What v_mfma_f32_4x4x1_16b_f32 does with per-lane operands (GPU box): prints, for every lane and result register, which lanes' A and B made it."""
import ctypes as C, os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = C.CDLL(os.path.join(root, "gpujpeg_amd", "lib", "libgj_testhooks.so"))
fn = h.gj_test_mfma4x4
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
l = np.arange(64)
a = (l + 1).astype(np.float32)
b = (1 + 64 * (l + 1)).astype(np.float32)
for via in (0, 1):
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    dc = torch.zeros(256, dtype=torch.float32, device="cuda")
    do = torch.zeros(256, dtype=torch.float32, device="cuda")
    assert fn(da.data_ptr(), db.data_ptr(), dc.data_ptr(), do.data_ptr(), via, None) == 0
    torch.cuda.synchronize()
    o = do.cpu().numpy().astype(np.int64).reshape(64, 4)
    x = (o - 1) % 64  # lane of A (a = x + 1)
    y = (o // (x + 1) - 1) // 64 - 1  # lane of B
    print("via_asm", via)
    for lane in (0, 1, 2, 3, 4, 5, 17, 63):
        print(" lane", lane, "regs:", [(int(x[lane, v]), int(y[lane, v])) for v in range(4)], "(A lane, B lane)")
    ok = all(x[q, v] == (q & ~3) + v and y[q, v] == q for q in range(64) for v in range(4))
    print(" assumed layout (D[v] of lane q = A[lane 4(q/4)+v] * B[lane q]):", ok)
    # C: does register v of lane q come back in register v of lane q?
    c = np.arange(256, dtype=np.float32) * 1000003.0
    dc = torch.from_numpy(c).cuda()
    assert fn(torch.zeros(64, device="cuda").data_ptr(), db.data_ptr(), dc.data_ptr(), do.data_ptr(), via, None) == 0
    torch.cuda.synchronize()
    print(" C passes through in place:", bool(np.array_equal(do.cpu().numpy(), c)))
