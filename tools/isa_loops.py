#!/usr/bin/env python3
"""Loop bodies of a gfx950 kernel by instruction class (an instruction-issue roofline that is counted, not assumed).

usage: tools/isa_loops.py <host object or code object> <kernel name substring> [--all]

Disassembles the kernel (llvm-objdump), finds the backward branches, and prints for every innermost loop the number of instructions per
class with the issue cycles measured for the class on this GPU (tools/ubench/valu_rate.hip, profiles/r2_09_ubench.txt: a wave64 vector
instruction occupies its SIMD for ~2.4 cycles when it is a plain add / sub / and / or / xor / mov / ashr / fp32 mul-add-fma, ~4.3 cycles
otherwise). Weighted by trip counts (which the caller knows) this gives the issue floor of the loop."""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_ashrrev_i32", "v_mul_f32", "v_add_f32",
        "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_not_b32"}
FAST_CYC, SLOW_CYC = 2.4, 4.3


def code_object(path):
    data = open(path, "rb").read()
    if data[:4] == b"\x7fELF" and b"__CLANG_OFFLOAD_BUNDLE__" not in data[:4096] and b".hip_fatbin" not in data:
        return path
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, "dev.co")
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={path}", f"--output={out}", "--unbundle"],
                       capture_output=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={out}", "--unbundle"])
    return out


def klass(m):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", m)
    if m.startswith("v_"):
        if m.endswith("_sdwa") or m.endswith("_dpp"):
            return "valu4"
        return "valu2" if base in FAST else "valu4"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt") or m.startswith("s_nop"):
        return "wait"
    if m.startswith("s_cbranch") or m.startswith("s_branch"):
        return "branch"
    if m.startswith("s_load") or m.startswith("s_buffer_load"):
        return "smem"
    return "salu"


def mix_json():
    """--mix: the share of 4-cycle-class vector instructions of the product's hot kernels, over their loop bodies (the code that runs per
    symbol / per token) and over the whole kernel, as JSON for bench.py's class-weighted issue floor"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kernels = {"enc:k_encode_rgb444": ("gj_enc_tiles.o", "k_encode_rgb444<1, 3, false>"), "enc:k_gather": ("gj_enc_assemble.o", "k_gather"), "enc:k_assemble": ("gj_enc_assemble.o", "k_assemble"),
               "enc:k_encode_uyvy422": ("gj_enc_tiles.o", "k_encode_uyvy422"),
               "dec:k_huffman_decode_tok": ("gj_dec_entropy_tok.o", "k_huffman_decode_tok<true>"), "dec:k_idct_tok_rgb444": ("gj_dec_idct.o", "k_idct_tok_rgb444<3, 1>"),
               "dec:k_huffman_decode_win": ("gj_dec_entropy_seq.o", "k_huffman_decode_win<true>"), "dec:k_huffman_decode_seq": ("gj_dec_entropy_seq.o", "k_huffman_decode_seq<true, false>"), "dec:k_idct_tok_uyvy422": ("gj_dec_idct.o", "k_idct_tok_uyvy422"),
               "dec:k_huffman_decode_par": ("gj_dec_entropy_par.o", "k_huffman_decode_par<false, 16>"), "dec:k_idct_fused_rgb444": ("gj_dec_idct.o", "k_idct_fused_rgb444<3, 1>")}
    out = {}
    for key, (obj, name) in kernels.items():
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", code_object(os.path.join(root, "build", "obj", obj))], capture_output=True, text=True).stdout
        cur, body = None, []
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
            if m:
                cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
            if m and cur == name:
                body.append((int(m.group(3), 16), m.group(1), m.group(2)))
        if not body:
            continue
        idx = {a: i for i, (a, _, _) in enumerate(body)}
        inloop = [False] * len(body)
        for i, (a, mn, ops) in enumerate(body):
            if mn.startswith(("s_cbranch", "s_branch")):
                try:
                    off = int(ops.split()[0])
                except (ValueError, IndexError):
                    continue
                off = off - 65536 if off >= 32768 else off
                tgt = a + 4 + 4 * off
                if tgt <= a and tgt in idx:
                    for q in range(idx[tgt], i + 1):
                        inloop[q] = True
        def share(sel):
            v2 = sum(1 for (x, f) in zip(body, sel) if f and klass(x[1]) == "valu2")
            v4 = sum(1 for (x, f) in zip(body, sel) if f and klass(x[1]) == "valu4")
            return {"valu2": v2, "valu4": v4, "share4": round(v4 / max(1, v2 + v4), 3)}
        out[key] = {"kernel": name, "loops": share(inloop), "whole": share([True] * len(body))}
    print(json.dumps({"note": "static instruction mix (tools/isa_loops.py --mix): vector instructions of the 2.4-cycle class (add / sub / and / or / xor / mov / not / "
                              "ashr / fp32 mul-add-fma) and of the 4.3-cycle class (everything else) as issue rates were measured on this GPU "
                              "(profiles/r2_09_ubench.txt); `loops` = instructions inside loops (what runs per symbol / token), `whole` = the kernel's whole text",
                      "cycles": {"valu2": FAST_CYC, "valu4": SLOW_CYC}, "kernels": out}, indent=1))


def main():
    if "--mix" in sys.argv:
        return mix_json()
    path, name = sys.argv[1], sys.argv[2]
    show_all = "--all" in sys.argv
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", code_object(path)], capture_output=True, text=True).stdout
    cur, kernels = None, {}
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m and cur:
            kernels[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    for kname, ins in kernels.items():
        dem = subprocess.run(["c++filt", kname], capture_output=True, text=True).stdout.strip()
        if name not in dem and name not in kname:
            continue
        print(f"== {dem.split('(')[0]}: {len(ins)} instructions")
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (a, m, ops) in enumerate(ins):
            if m.startswith(("s_cbranch", "s_branch")):
                try:
                    off = int(ops.split()[0])
                except (ValueError, IndexError):
                    continue
                if off >= 32768:
                    off -= 65536
                tgt = a + 4 + 4 * off
                if tgt <= a and tgt in addr_index:
                    loops.append((addr_index[tgt], i))
        inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
        for lo, hi in sorted(loops if show_all else inner):
            body = ins[lo:hi + 1]
            cnt = {}
            for _, m, _ in body:
                cnt[klass(m)] = cnt.get(klass(m), 0) + 1
            v2, v4 = cnt.get("valu2", 0), cnt.get("valu4", 0)
            tag = "inner" if (lo, hi) in inner else "outer"
            print(f"  loop @{ins[lo][0]:#x}..{ins[hi][0]:#x} ({tag}): {len(body)} instr; VALU {v2 + v4} (2-cycle class {v2}, 4-cycle class {v4}) = "
                  f"{v2 * FAST_CYC + v4 * SLOW_CYC:.0f} issue cycles; LDS {cnt.get('lds', 0)}, VMEM {cnt.get('vmem', 0)}, SALU {cnt.get('salu', 0)}, "
                  f"branch {cnt.get('branch', 0)}, wait {cnt.get('wait', 0)}")
            if "--dump" in sys.argv:
                for a, m, ops in body:
                    print(f"      {a:#x}  {m} {ops}")


if __name__ == "__main__":
    main()
