#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X JPEG hot path.

metric: Mpix/s encode+decode (8K RGB q75), BASELINE.json. One step = one pass of the hot path over one
synthetic 7680x4320 RGB frame that is already resident in HBM: gpujpeg_encoder_encode (GPU_IMAGE input,
JPEG left in HBM) followed by gpujpeg_decoder_decode of that JPEG (device-resident stream, pixels written to
HBM), both through the libgpujpeg C ABI. `value` = pixels of all ranks / max-over-ranks time of K steps.

Multi-GPU: frames are independent, so ranks shard the frame batch with no data-path collective (weak scaling,
one frame per rank and step). N > 1 is launched by torch.distributed.run; RCCL is used only for the barrier
and the max-over-ranks reduction of the timing.

Extra objects on the JSON line:
  roofline      dominant kernel of the step (largest average hipEvent duration), algorithmic bytes
                (raw RGB in + JPEG out for encoder kernels, JPEG in + raw RGB out for decoder kernels;
                SURVEY.md 8d) divided by that duration, against 8 TB/s HBM3E
  cpu_baseline  the reference's own host C code + the restated CUDA-only stages (oracle/_ref, kind
                "reference"; falls back to the pure restatement, kind "port") timed on one host core on a
                bounded sample of the same workload
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); with the default stream and four pipeline
# streams two of them would share a queue and serialise. Must be set before the HIP runtime starts (measured: 4 pipelines on 8
# queues 124.5 Gpix/s, on 4 queues 105.6; 3 pipelines 119.1 either way).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

WORKLOADS = {
    "hd": (1920, 1080), "4k": (3840, 2160), "8k": (7680, 4320), "16k": (15360, 8640),
    # BASELINE.json config 4: 16K YCbCr 4:2:2 (UYVY) interleaved q90 (not the headline; same measurement)
    "16k422": (15360, 8640), "8k422": (7680, 4320), "hd422": (1920, 1080),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E
# HBM bytes per launch of the dominant kernels from the PMC passes committed under profiles/ (8K RGB q75 natural frame):
# FETCH_SIZE doubled (gfx950 counts 128 B requests as 64 B, guide section HBM) + WRITE_SIZE
TRAFFIC_BYTES = {"enc:k_encode_rgb444": int((49802.4 * 2 + 8350.1) * 1024), "dec:k_huffman_decode_par": int((4463.7 * 2 + 70987.8) * 1024),
                 "dec:k_idct_tok_rgb444": int((26587.8 * 2 + 103275.0) * 1024)}  # profiles/r1_06_solo_hbm_traffic.txt (token mode)


def synth_frame(width, height, pattern, seed, device):
    """Deterministic synthetic RGB frame generated on the device.
    natural: smooth structure + texture + mild sensor noise (compresses like a photograph at q75)
    noise:   the reference's `.tst` LCG noise semantics, worst case for the entropy coder
    gradient: the reference's `.tst` default, best case"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if pattern == "noise":
        return torch.randint(0, 256, (height, width, 3), dtype=torch.uint8, device=device, generator=g)
    yy = torch.arange(height, device=device, dtype=torch.float32).view(-1, 1)
    xx = torch.arange(width, device=device, dtype=torch.float32).view(1, -1)
    if pattern == "gradient":
        row = (torch.arange(height, device=device) * 255 // height).to(torch.uint8).view(-1, 1, 1)
        return row.expand(height, width, 3).contiguous()
    chans = []
    for k, (fx, fy, ph) in enumerate([(1 / 97.0, 1 / 61.0, 0.3), (1 / 53.0, 1 / 131.0, 1.1), (1 / 211.0, 1 / 89.0, 2.0)]):
        base = 128 + 70 * torch.sin(xx * fx + ph) * torch.cos(yy * fy) + 30 * torch.sin((xx + yy) * fx * 3.1 + k)
        tex = 12 * torch.sin(xx * 0.9 + yy * 0.35 + k) * torch.sin(yy * 0.7 - xx * 0.11)
        nz = 3.0 * torch.randn((height, width), device=device, generator=g)
        chans.append(base + tex + nz)
    return torch.stack(chans, -1).clamp(0, 255).to(torch.uint8).contiguous()


def cpu_baseline(width, height, frame_host, seconds_budget=20.0, is422=False, quality=75):
    """Reference CPU path on the host cores (1 thread): encode + decode of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    kind = "reference" if O.have_ref() else "port"
    t_start = time.time()
    frames = 0
    if kind == "reference":
        ref = G.Library(O.REF_PATH)
        enc, dec = G.Encoder(ref), G.Decoder(ref)
        p = ref.default_parameters()
        p.restart_interval, p.verbose = G.RESTART_AUTO, -1
        pi = ref.default_image_parameters()
        pi.width, pi.height = width, height
        p.quality = quality
        if is422:
            pi.pixel_format, pi.color_space = G.P1020_422, G.YCBCR_JPEG
            p.interleaved = 1
            dec.set_output_format(G.YCBCR_JPEG, G.P1020_422)
        while True:
            jpeg = enc.encode(p, pi, frame_host)
            dec.decode(jpeg)
            frames += 1
            if time.time() - t_start > seconds_budget * 0.5 or frames >= 4:
                break
    else:
        img = (O.make_image(width, height, pixel_format=3, color_space=3, quality=quality, interleaved=1) if is422
               else O.make_image(width, height, quality=quality))
        while True:
            jpeg = O.encode(img, frame_host)
            O.decode(jpeg, 3, 3) if is422 else O.decode(jpeg)
            frames += 1
            if time.time() - t_start > seconds_budget * 0.5 or frames >= 4:
                break
    dt = time.time() - t_start
    return {"value": round(width * height * frames / dt / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": kind,
            "sample": f"{frames} x encode+decode of the {width}x{height} {'UYVY 4:2:2' if is422 else 'RGB'} q{quality} frame, single thread, "
                      f"reference host C (writer/reader/CPU Huffman) + restated colour/DCT/IDCT stages, gcc -O2"}


_CPU_CHILD = r"""
import sys, time
import numpy as np
root, path, w, h, q, is422, frames = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6] == "1", int(sys.argv[7])
sys.path.insert(0, root); sys.path.insert(0, root + "/oracle")
import oracle as O
from gpujpeg_amd import libgpujpeg as G
frame = np.load(path)
ref = G.Library(O.REF_PATH) if O.have_ref() else None
t0 = time.time()
if ref is not None:
    enc, dec = G.Encoder(ref), G.Decoder(ref)
    p = ref.default_parameters(); p.restart_interval, p.verbose, p.quality = G.RESTART_AUTO, -1, q
    pi = ref.default_image_parameters(); pi.width, pi.height = w, h
    if is422:
        pi.pixel_format, pi.color_space = G.P1020_422, G.YCBCR_JPEG; p.interleaved = 1; dec.set_output_format(G.YCBCR_JPEG, G.P1020_422)
    for _ in range(frames):
        dec.decode(enc.encode(p, pi, frame))
else:
    img = O.make_image(w, h, pixel_format=3, color_space=3, quality=q, interleaved=1) if is422 else O.make_image(w, h, quality=q)
    for _ in range(frames):
        j = O.encode(img, frame); O.decode(j, 3, 3) if is422 else O.decode(j)
print(time.time() - t0)
"""


def cpu_baseline_all_cores(width, height, frame_host, is422=False, quality=75, max_procs=32):
    """the reference's CPU path admits frame-level parallelism only -- one independent encode+decode per
    process on as many host cores as there are (capped), the same frame in each; aggregate Mpix/s over the slowest process."""
    import subprocess
    import tempfile
    procs = max(1, min(max_procs, (os.cpu_count() or 1)))
    with tempfile.NamedTemporaryFile(suffix=".npy", dir="/dev/shm" if os.path.isdir("/dev/shm") else None, delete=False) as f:
        np.save(f, frame_host)
        path = f.name
    try:
        t0 = time.time()
        kids = [subprocess.Popen([sys.executable, "-c", _CPU_CHILD, ROOT, path, str(width), str(height), str(quality), "1" if is422 else "0", "1"],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(procs)]
        inner = [float(k.communicate(timeout=300)[0].decode().strip().splitlines()[-1]) for k in kids]
        wall = time.time() - t0
    finally:
        os.unlink(path)
    return {"value": round(width * height * procs / max(inner) / 1e6, 3), "unit": "Mpix/s", "cores": procs, "kind": "reference" if os.path.exists(
        os.path.join(ROOT, "oracle", "_ref", "libgpujpeg_ref.so")) else "port",
        "sample": f"{procs} processes x 1 encode+decode of the same {width}x{height} frame at once (frame-level parallelism, the only kind the "
                  f"reference's CPU code admits); slowest process {max(inner):.2f} s, {wall:.1f} s including start-up"}


def run_batch(args, lib, device, local_rank, rank, world, width, height):
    """BASELINE.json config 5 / SURVEY.md 8(d): a batch of independent frames, frame i seeded 12345 + i, sharded over the
    ranks by gpujpeg_amd.sharding.shard_frames (static round-robin, no data-path collective). Every rank keeps its shard
    resident in HBM, S pipelines (stream + encoder + decoder + host thread) walk disjoint slices of it."""
    import threading
    import torch.distributed as dist
    from gpujpeg_amd.sharding import barrier_and_max, gather_counts, shard_frames
    if args.workload.endswith("422"):
        raise SystemExit("--batch is defined for the RGB workloads")
    mine = shard_frames(args.batch, rank, world)
    frames = [synth_frame(width, height, args.pattern, 12345 + i, device) for i in mine]
    S = max(1, min(args.streams, len(frames)))
    p = lib.default_parameters()
    p.quality, p.restart_interval, p.verbose = args.quality, G.RESTART_AUTO, -1
    pi = lib.default_image_parameters()
    pi.width, pi.height = width, height
    lanes = []
    for si in range(S):
        ts = torch.cuda.Stream(device)
        e, d = G.Encoder(lib, ts.cuda_stream), G.Decoder(lib, ts.cuda_stream)
        assert e.set_option("enc_opt_out", "enc_out_val_device") == 0
        lanes.append({"frames": frames[si::S], "out": torch.empty_like(frames[0]), "enc": e, "dec": d, "bytes": 0})
    torch.cuda.synchronize()

    def one_pass(ln):
        nbytes = 0
        for f in ln["frames"]:
            jp, js = ln["enc"].encode_noclone(p, pi, f.data_ptr(), gpu=True)
            o = G.DecoderOutput()
            o.type, o.data = G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, ln["out"].data_ptr()
            assert lib.L.gpujpeg_decoder_decode(ln["dec"].h, C.cast(jp, C.c_void_p), js, C.byref(o)) == 0
            nbytes += js
        ln["bytes"] = nbytes

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    go = threading.Event()

    def worker(idx, passes, wait):
        torch.cuda.set_device(local_rank)
        if wait:
            go.wait()
        for _ in range(passes):
            one_pass(lanes[idx])

    for passes, timed in ((args.warmup, False), (args.steps, True)):
        threads = [threading.Thread(target=worker, args=(i, passes, timed)) for i in range(S)]
        for t in threads:
            t.start()
        if timed:
            barrier()
            t0 = time.perf_counter()
            go.set()
        for t in threads:
            t.join()
        if timed:
            barrier()
            elapsed = time.perf_counter() - t0
    # the last decoded frame of pipeline 0 must match its input closely (sanity of the timed work, not the parity test)
    last = lanes[0]["frames"][-1].float()
    mse = float(((lanes[0]["out"].float() - last) ** 2).mean().item())
    elapsed = barrier_and_max(elapsed, device)
    total = gather_counts(len(mine), device)
    jpeg_bytes = gather_counts(sum(ln["bytes"] for ln in lanes), device)
    for ln in lanes:
        ln["enc"].close()
        ln["dec"].close()
    if rank == 0:
        fps = total * args.steps / elapsed
        print(json.dumps({
            "metric": f"frames/s encode+decode (batch of {args.batch} {args.workload} frames)", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "dtype_detail": "u8 samples, fp32 colour transform and DCT (bit-exact with the reference's integer / float arithmetic), i16 coefficients",
            "data": f"synthetic ({args.pattern}), {args.batch} distinct {width}x{height} frames (seed 12345 + i) resident in HBM, sharded round-robin",
            "config": {"workload": f"{args.batch} x {width}x{height} RGB 4:4:4 q{args.quality} non-interleaved, restart auto, encode then decode of "
                                   f"every frame per step", "frames_total": total, "frames_per_gpu": len(mine), "streams_per_gpu": S,
                       "jpeg_bytes_total": jpeg_bytes, "parallelism": f"frame-sharded x{world}, no collective"},
            "mpix_s": round(fps * width * height / 1e6, 2), "psnr_last_frame_db": round(10 * np.log10(255.0 ** 2 / max(mse, 1e-9)), 2)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="8k", choices=sorted(WORKLOADS))
    ap.add_argument("--pattern", default="natural", choices=["natural", "noise", "gradient"])
    ap.add_argument("--quality", type=int, default=75)
    ap.add_argument("--batch", type=int, default=0,
                    help="BASELINE.json config 5: a fixed batch of this many distinct frames (seeds 12345 + i) sharded over the ranks "
                         "(strong scaling, frames/s); a step is one pass over the whole batch")
    ap.add_argument("--streams", type=int, default=4, help="independent encoder+decoder pairs per GPU, each on its own HIP stream and host thread; "
                    "one step codes one frame per stream (hides the host side of one call behind the kernels of the other)")
    ap.add_argument("--mode", default="both", choices=["both", "encode", "decode"],
                    help="what a timed step does: encode then decode (the headline metric), or only one direction (SURVEY 8d asks for both "
                         "separately; the decoder then decodes the stream of the warm-up's last encode again and again)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU path on all host cores (one frame per process), SURVEY 8(d) row (ii)")
    ap.add_argument("--internal-rgb", action="store_true", help="code RGB without colour transform (tuning aid; not the headline config)")
    ap.add_argument("--calibrate", action="store_true", help="run a 256 MiB device fill + copy first (known byte counts for calibrating PMC traffic counters)")
    ap.add_argument("--keep-coefs", action="store_true", help="decoder keeps its coefficients in HBM (adds the per-frame clear; tuning aid)")
    ap.add_argument("--verify", action="store_true", help="check the round trip of the last frame against the oracle (slow)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    lib = G.Library()  # raises if the HIP library has not been built: there is no fallback
    assert lib.L.gpujpeg_init_device(local_rank, 0) == 0
    width, height = WORKLOADS[args.workload]
    is422 = args.workload.endswith("422")
    if args.batch:
        return run_batch(args, lib, device, local_rank, rank, world, width, height)
    frame = synth_frame(width, height, args.pattern, 12345 + rank, device)
    if is422:  # packed UYVY from the synthetic RGB frame: channels reused as Y / Cb / Cr, chroma point-sampled
        f = frame.view(height, width, 3)
        uyvy = torch.empty((height, width, 2), dtype=torch.uint8, device=device)
        uyvy[:, :, 1] = f[:, :, 0]
        uyvy[:, 0::2, 0] = f[:, 0::2, 1] // 2 + 64
        uyvy[:, 1::2, 0] = f[:, 0::2, 2] // 2 + 64
        frame = uyvy.contiguous()
        if args.quality == 75:
            args.quality = 90
    import threading
    S = max(1, args.streams)
    lanes = []  # one pipeline per stream: its own frame, output buffer, stream, encoder and decoder
    for si in range(S):
        f = frame if si == 0 else frame.clone()
        ts = torch.cuda.current_stream(device) if S == 1 else torch.cuda.Stream(device)
        e, d = G.Encoder(lib, ts.cuda_stream), G.Decoder(lib, ts.cuda_stream)
        assert e.set_option("enc_opt_out", "enc_out_val_device") == 0
        lanes.append({"frame": f, "out": torch.empty_like(f), "stream": ts, "enc": e, "dec": d})
    torch.cuda.synchronize()
    enc, dec, out = lanes[0]["enc"], lanes[0]["dec"], lanes[0]["out"]
    p = lib.default_parameters()
    p.quality, p.restart_interval, p.verbose, p.perf_stats = args.quality, G.RESTART_AUTO, -1, 1
    if args.internal_rgb:
        p.color_space_internal = 1
    pi = lib.default_image_parameters()
    pi.width, pi.height = width, height
    if is422:
        pi.pixel_format, pi.color_space = G.P1020_422, G.YCBCR_JPEG
        p.interleaved = 1
        lib.L.gpujpeg_parameters_chroma_subsampling(C.byref(p), G.SUBSAMPLING_422)
    for ln in lanes:
        if args.keep_coefs:
            ln["dec"].keep_coefficients()
        ln["dec"].init(p, lib.default_image_parameters())  # turns perf_stats on for the decoder (same API as the reference)
        if is422:
            ln["dec"].set_output_format(G.YCBCR_JPEG, G.P1020_422)

    def step(ln):
        jptr, jsize = ln["enc"].encode_noclone(p, pi, ln["frame"].data_ptr(), gpu=True)
        o = G.DecoderOutput()
        o.type, o.data = G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, ln["out"].data_ptr()
        rc = lib.L.gpujpeg_decoder_decode(ln["dec"].h, C.cast(jptr, C.c_void_p), jsize, C.byref(o))
        assert rc == 0
        return jptr, jsize

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.calibrate:
        cal_a = torch.zeros(2 ** 26, dtype=torch.int32, device=device)
        cal_b = cal_a.clone()
        torch.cuda.synchronize()
        del cal_a, cal_b
    solo_ms = np.zeros(8)  # kernel durations with the GPU to themselves (untimed warm-up of pipeline 0; reference for the roofline)
    for _ in range(max(1, args.warmup)):
        for ln in lanes:
            jptr, jsize = step(ln)
            ln["last"] = (jptr, jsize)
    torch.cuda.synchronize()
    for _ in range(3):
        jptr, jsize = step(lanes[0])
        torch.cuda.synchronize()
        solo_ms += np.array(list(lanes[0]["enc"].kernel_times()) + list(lanes[0]["dec"].kernel_times()))
    solo_ms /= 3
    enc_ms = np.zeros(5)
    dec_ms = np.zeros(3)
    walls = [0.0, 0.0]
    go = threading.Event()

    def worker(idx):
        torch.cuda.set_device(local_rank)  # the HIP device is per host thread
        ln = lanes[idx]
        go.wait()
        jp, js = ln["last"]
        for _ in range(args.steps):
            a = time.perf_counter()
            if args.mode != "decode":
                jp, js = ln["enc"].encode_noclone(p, pi, ln["frame"].data_ptr(), gpu=True)
            b = time.perf_counter()
            if args.mode != "encode":
                o = G.DecoderOutput()
                o.type, o.data = G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, ln["out"].data_ptr()
                assert lib.L.gpujpeg_decoder_decode(ln["dec"].h, C.cast(jp, C.c_void_p), js, C.byref(o)) == 0
            c = time.perf_counter()
            if idx == 0:  # per-kernel hipEvent durations of pipeline 0 (its kernels may share the GPU with the other pipelines')
                walls[0] += b - a
                walls[1] += c - b
                if args.mode != "decode":
                    enc_ms[:] += np.array(ln["enc"].kernel_times())
                if args.mode != "encode":
                    dec_ms[:] += np.array(ln["dec"].kernel_times())
            ln["last"] = (jp, js)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
    for t in threads:
        t.start()
    barrier()
    t0 = time.perf_counter()
    go.set()
    for t in threads:
        t.join()
    barrier()
    elapsed = time.perf_counter() - t0
    enc_wall, dec_wall = walls
    jptr, jsize = lanes[0]["last"]
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    enc_ms /= args.steps
    dec_ms /= args.steps

    result = None
    if rank == 0:
        pixels = width * height
        raw_bytes = pixels * (2 if is422 else 3)
        whole = enc_ms[1] < 0.02  # fully fused encoder: pixels -> segment streams in one kernel (event slots 0/1 are empty)
        fmt = "uyvy422" if is422 else "rgb444"
        # the decoder picks token mode for large, sparse frames (gj_decode.hip: >= 900 K blocks, <= 8 stream bytes per block)
        nblocks = ((width + 7) // 8) * ((height + 7) // 8) * (2 if is422 else 3)
        token_mode = (not is422) and nblocks >= 900000 and jsize <= 8 * nblocks and not args.keep_coefs \
            and not os.environ.get("GJ_DEC_NO_TOKENS")
        names = ["enc:k_preprocess", f"enc:k_fused_{fmt}(pre+dct+quant)",
                 f"enc:k_encode_{fmt}(pixels->entropy-coded segments)" if whole else "enc:k_huffman", "enc:k_scan_segments", "enc:k_assemble",
                 "dec:k_huffman_decode_par(+fallback launch)",
                 f"dec:k_idct_{'tok' if token_mode else 'fused'}_{fmt}(idct+post)", "dec:k_postprocess"]
        # the PMC passes under profiles/ were taken on the default 8K workload only
        traffic_known = args.workload == "8k" and args.quality == 75 and args.pattern == "natural"
        durs = list(enc_ms) + list(dec_ms)
        dom = int(np.argmax(durs))
        alg = raw_bytes + jsize  # encoder: raw in + JPEG out; decoder: JPEG in + raw out (same sum)
        achieved = alg / (durs[dom] * 1e-3) / 1e9
        result = {
            "metric": ("Mpix/s encode+decode (8K RGB q75)" if args.workload == "8k" else f"Mpix/s encode+decode ({args.workload})") if args.mode == "both"
                      else f"Mpix/s {args.mode} only ({args.workload})", "value": round(pixels * world * S * args.steps / elapsed / 1e6, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_detail": "u8 samples, fp32 colour transform and DCT (bit-exact with the reference's integer / float arithmetic), i16 coefficients",
            "data": f"synthetic ({args.pattern}), {S} {width}x{height} frame(s) per rank resident in HBM, one per stream",
            "config": {"workload": (f"{width}x{height} YCbCr 4:2:2 (UYVY) q{args.quality} interleaved, restart auto, encode then decode per step" if is422 else
                                    f"{width}x{height} RGB 4:4:4 q{args.quality} non-interleaved, restart auto ({width}x{height} -> "
                                    f"{'36' if args.workload in ('8k', '16k') else 'auto'}), encode then decode per step"),
                       "frames_per_step_per_gpu": S, "streams_per_gpu": S, "jpeg_bytes": int(jsize), "parallelism": f"frame-sharded x{world}, no collective"},
            # API calls of pipeline 0 alone (one call at a time per pipeline; the aggregate of all pipelines is `value`)
            "encode_mpix_s": round(pixels * args.steps / enc_wall / 1e6, 2) if args.mode != "decode" else None,
            "decode_mpix_s": round(pixels * args.steps / dec_wall / 1e6, 2) if args.mode != "encode" else None,
            "kernel_ms": {n: round(float(d), 4) for n, d in zip(names, durs)},
            "gpu_only_ms": {"encode": round(float(enc_ms.sum()), 4), "decode": round(float(dec_ms.sum()), 4)},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": TRAFFIC_BYTES.get(names[dom].split("(")[0]) if traffic_known else None,
                         "algorithmic_bytes_per_launch": int(alg), "concurrent_pipelines": S,
                         "note": "duration = average hipEvent duration of one launch in the timed region, where launches of the other "
                                 "pipelines share the GPU; traffic = FETCH_SIZE x 2 + WRITE_SIZE per launch from profiles/ (separate PMC passes)"},
            "roofline_solo": {"kernel": names[int(np.argmax(solo_ms))], "ms": round(float(solo_ms.max()), 4),
                              "achieved": round(alg / (float(solo_ms.max()) * 1e-3) / 1e9, 2), "unit": "GB/s",
                              "frac": round(alg / (float(solo_ms.max()) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              "kernel_ms": {n: round(float(d), 4) for n, d in zip(names, solo_ms)}},
        }
        if args.verify:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            host = frame.cpu().numpy().reshape(-1)
            img = (O.make_image(width, height, pixel_format=3, color_space=3, quality=args.quality, interleaved=1) if is422
                   else O.make_image(width, height, quality=args.quality, color_space_internal=1 if args.internal_rgb else 3))
            want = O.encode(img, host)
            got = np.ctypeslib.as_array(C.cast(jptr, C.POINTER(C.c_uint8)), shape=(1,))  # device pointer: copy through torch
            jt = torch.empty(jsize, dtype=torch.uint8, device=device)
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(jt.data_ptr()), C.cast(jptr, C.c_void_p), C.c_size_t(jsize), 3)
            torch.cuda.synchronize()
            result["verified_bit_exact_encode"] = bool(np.array_equal(jt.cpu().numpy(), want))
            result["verified_bit_exact_decode"] = bool(np.array_equal(out.cpu().numpy().reshape(-1), (O.decode(want, 3, 3) if is422 else O.decode(want))[0]))
            del got
        if not args.no_cpu_baseline and world == 1:  # reported baseline, on the host cores of rank 0 at N = 1 only
            host_frame = frame.cpu().numpy().reshape(-1)
            result["cpu_baseline"] = cpu_baseline(width, height, host_frame, is422=is422, quality=args.quality)
            if args.cpu_all_cores:
                result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(width, height, host_frame, is422=is422, quality=args.quality)
        print(json.dumps(result), flush=True)
    for ln in lanes:
        ln["enc"].close()
        ln["dec"].close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
