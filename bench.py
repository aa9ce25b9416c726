#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X JPEG hot path through the libgpujpeg C ABI.

Headline (BASELINE.json `metric`): Mpix/s encode+decode of 8K RGB 4:4:4 q75 frames, non-interleaved, restart interval auto (36),
frames resident in HBM when the timed region starts. One STEP = one pass of the hot path over one batch of synthetic input:
every one of the S pipelines (HIP stream + encoder + decoder + host thread) codes and decodes `frames_per_step_per_pipeline`
frames -- gpujpeg_encoder_encode (GPU_IMAGE input, stream left in HBM) followed by gpujpeg_decoder_decode of that stream into
HBM. The batch size is fixed during warm-up so that the K timed steps take at least half a second whatever K is; `value` =
pixels of all ranks / max-over-ranks time of exactly K steps, `ms_per_step` = that time / K.

Multi-GPU: frames are independent, ranks shard them with no data-path collective (weak scaling; RCCL carries the barrier and
the max-over-ranks of the elapsed time only). N > 1 is launched by torch.distributed.run.

On rank 0 at N = 1 the same JSON line also carries (each a short bounded run; --lean skips them):
  roofline          the dominant kernel of the step: algorithmic bytes per launch (raw in + JPEG out for the encoder direction,
                    JPEG in + raw out for the decoder) / its average hipEvent duration in a SOLO timed region (one pipeline, the GPU
                    otherwise idle, events on the coder's own stream) against 8 TB/s; `by_kernel` has every kernel of the step,
                    `contended` the same kernel inside the headline region where four pipelines share the GPU; `traffic` = HBM bytes
                    per launch from the PMC passes under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE, profiles/r6_traffic.json; dropped when the
                    device sources differ from the ones profiled)
  encode_only / decode_only   each direction alone, device resident ("w/o PCIe" in the reference's tables)
  full_api          host buffers in and out (pinned), i.e. what a drop-in caller of the reference API sees, PCIe included
  workloads         HD / 4K / 8K / 16K RGB, 16K 4:2:2 interleaved q90 (BASELINE config 4), each with its own roofline; the 8K frame with the
                    reference's own contents: `8k_noise` (7680x4320.random_12345.tst), `8k_gradient` (7680x4320.gradient.tst), `8k_camera`
                    (its camera sample, tests/golden/make_camera_fixture.py) and `8k_camera_quality_sweep` (q10 ... q100: the sweep the reference's
                    README publishes); 256 x 4K batch (config 5) resident in HBM (`batch256_4k`) and
                    from pinned host memory in and out (`batch256_4k_host`); the same batch, 256 HD frames and 256 HD packed 4:2:2 frames through the
                    batch calls of include/gpujpeg_amd_ext.h -- every kernel launched once per chunk of frames -- (`batch256_4k_batched`,
                    `batch256_hd_batched`, `batch256_hd422_batched`) next to 256 HD frames one call per frame (`batch256_hd`)
The timed regions run with perf_stats = 0 (no per-kernel events); per-kernel durations come from their own short regions. The launch
threads are bound to idle cores of the GPU's NUMA node when several ranks share the node (--pin on / off forces or forbids it).
  cpu_baseline      the reference's CPU path as it exists (its host C with the CPU Huffman coders) + the restated scalar stages for
                    what it only has as CUDA, one thread, on a bounded sample; `idct_cpu_s` = its own gpujpeg_idct_cpu on the same frame
  cpu_baseline_all_cores      the same with one frame per process on the host's cores (-O3 -march=native build)
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

# HIP maps the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); with the default stream and four pipeline
# streams two of them would share a queue and serialise. Must be set before the HIP runtime starts (measured in round 1: 4 pipelines
# on 8 queues 124.5 Gpix/s, on 4 queues 105.6).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from gpujpeg_amd import libgpujpeg as G  # noqa: E402

WORKLOADS = {
    "hd": (1920, 1080), "4k": (3840, 2160), "8k": (7680, 4320), "16k": (15360, 8640),
    # BASELINE.json config 4: 16K YCbCr 4:2:2 (UYVY) interleaved q90
    "16k422": (15360, 8640), "8k422": (7680, 4320), "hd422": (1920, 1080),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E
DTYPE_DETAIL = "u8 samples, fp32 colour transform and DCT (bit-exact with the reference's integer / float arithmetic), i16 coefficients"


TRAFFIC_FILE = "r6_traffic.json"
ROTATE = 3  # distinct frames (and output buffers) per pipeline, see Spec


from gpujpeg_amd.source_hash import kernel_source_hash  # noqa: E402


def load_traffic(key="kernels", workload="8k"):
    """HBM bytes per launch (`kernels`) / vector instructions per launch (`valu_insts`) of a workload from the committed PMC passes
    (profiles/r6_traffic.json, written by tools/profile.sh); empty when the device sources have changed since they were taken."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
        for w in d.get("workloads", {}).values():  # bench.py times the two marker kernels with one pair of events: one name for their sum
            for table in ("kernels", "valu_insts"):
                t = w.get(table, {})
                if "dec:k_marker_scan" in t and "dec:k_marker_table" in t:
                    t["dec:k_markers"] = t["dec:k_marker_scan"] + t["dec:k_marker_table"]
        if d.get("source_hash") != kernel_source_hash():
            return {}
        return d.get("workloads", {}).get(workload, {}).get(key, {})
    except Exception:
        return {}


# The kernels of this path are bound by vector-instruction issue, not by HBM. A wave64 instruction occupies its SIMD for ~2.4
# cycles when it is a plain add / sub / and / or / xor / mov / not / ashr / fp32 mul-add-fma and for ~4.3 cycles otherwise (shifts, bit-field,
# compare, select, convert, packed-fp32, three-operand integer: tools/ubench/valu_rate.hip on this GPU, profiles/r2_09_ubench.txt). The
# issue floor of a kernel = SQ_INSTS_VALU per launch (PMC pass) x the cycles of its class mix (static mix of the kernel's code,
# tools/isa_loops.py --mix -> profiles/r6_isa_mix.json) / 1024 SIMDs / 2.4 GHz; the floors with every instruction in the fast and in
# the slow class bracket it.
SIMDS, CLOCK_HZ = 1024, 2.4e9


def load_isa_mix():
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r6_isa_mix.json")))
        return d["cycles"], {k: v["whole"]["share4"] for k, v in d["kernels"].items()}
    except Exception:
        return {"valu2": 2.4, "valu4": 4.3}, {}


def synth_frame(lib, width, height, pattern, seed, device):
    """Deterministic synthetic RGB frame in HBM.
    natural:  smooth structure + texture + mild sensor noise, generated on the device (compresses like a photograph at q75)
    noise / gradient: the reference's own `.tst` generators (src/utils/image_delegate.c:562-603: the 1664525 / 1013904223 LCG
              with seed `seed`, and rows of i * 255 / H) through gpujpeg_image_load_from_file, so the numbers can be reproduced
              with `gpujpegtool WxH.random_<seed>.tst`"""
    if pattern == "camera":
        return camera_frame(lib, width, height, device)
    if pattern in ("noise", "gradient"):
        name = f"{width}x{height}.{'random_%d' % seed if pattern == 'noise' else 'gradient'}.tst".encode()
        img, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
        lib.L.gpujpeg_image_load_from_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
        assert lib.L.gpujpeg_image_load_from_file(name, C.byref(img), C.byref(size)) == 0
        host = np.ctypeslib.as_array(img, shape=(size.value,))
        t = torch.from_numpy(host.copy()).to(device).view(height, width, 3)
        lib.L.gpujpeg_image_destroy.argtypes = [C.POINTER(C.c_uint8)]
        lib.L.gpujpeg_image_destroy(img)
        return t
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    yy = torch.arange(height, device=device, dtype=torch.float32).view(-1, 1)
    xx = torch.arange(width, device=device, dtype=torch.float32).view(1, -1)
    chans = []
    for k, (fx, fy, ph) in enumerate([(1 / 97.0, 1 / 61.0, 0.3), (1 / 53.0, 1 / 131.0, 1.1), (1 / 211.0, 1 / 89.0, 2.0)]):
        base = 128 + 70 * torch.sin(xx * fx + ph) * torch.cos(yy * fy) + 30 * torch.sin((xx + yy) * fx * 3.1 + k)
        tex = 12 * torch.sin(xx * 0.9 + yy * 0.35 + k) * torch.sin(yy * 0.7 - xx * 0.11)
        nz = 3.0 * torch.randn((height, width), device=device, generator=g)
        chans.append((base + tex + nz).clamp(0, 255).to(torch.uint8))
    return torch.stack(chans, -1).contiguous()


DATA_NOTE = {"natural": "synthetic photograph-like frame generated on the device (synth_frame)",
             "noise": "the reference's {w}x{h}.random_12345.tst (src/utils/image_delegate.c:562-603)",
             "gradient": "the reference's {w}x{h}.gradient.tst (src/utils/image_delegate.c:562-603)",
             "camera": "the reference's camera sample colors/camera_bt709_422.yuv (HD), decoded to RGB once and tiled to {w}x{h}"}
CAMERA_FIXTURE = os.path.join(ROOT, "tests", "golden", "camera_bt709_422_q95.jpg")


def camera_frame(lib, width, height, device):
    """camera: the reference's camera sample (colors/camera_bt709_422.yuv, one HD frame; committed as a q95 JPEG by
    tests/golden/make_camera_fixture.py), decoded ONCE to RGB with the product's decoder and tiled to the size of the workload"""
    jpeg = np.fromfile(CAMERA_FIXTURE, np.uint8)
    d = G.Decoder(lib)
    px, pi = d.decode(jpeg)
    d.close()
    tile = torch.from_numpy(px.reshape(pi.height, pi.width, 3).copy()).to(device)
    reps = (-(-height // pi.height), -(-width // pi.width), 1)
    return tile.repeat(*reps)[:height, :width].contiguous()


def to_uyvy(frame):
    """packed UYVY from the synthetic RGB frame: channels reused as Y / Cb / Cr, chroma point-sampled"""
    h, w, _ = frame.shape
    uyvy = torch.empty((h, w, 2), dtype=torch.uint8, device=frame.device)
    uyvy[:, :, 1] = frame[:, :, 0]
    uyvy[:, 0::2, 0] = frame[:, 0::2, 1] // 2 + 64
    uyvy[:, 1::2, 0] = frame[:, 0::2, 2] // 2 + 64
    return uyvy.contiguous()


class Spec:
    """One workload: geometry, coding parameters, the frame."""

    def __init__(self, lib, name, pattern, quality, device, seed, internal_rgb=False):
        self.name, self.pattern = name, pattern
        self.width, self.height = WORKLOADS[name]
        self.is422 = name.endswith("422")
        self.quality = 90 if (self.is422 and quality == 75) else quality
        frame = synth_frame(lib, self.width, self.height, pattern, seed, device)
        self.frame = to_uyvy(frame) if self.is422 else frame
        # (VERDICT r5 #9a) every pipeline walks ROTATE distinct frames at distinct addresses (and as many output buffers), in the timed regions and in
        # the solo kernel timings: one 8K frame (99.5 MB) would sit in the 256 MB Infinity Cache between two launches, three do not. The natural
        # pattern differs by its noise seed; the reference's own contents (`.tst` noise / gradient, the camera sample) are the same file at other addresses
        self.variants = [self.frame]
        for k in range(1, ROTATE):
            if pattern == "natural":
                v = synth_frame(lib, self.width, self.height, pattern, seed + 1000 * k, device)
                self.variants.append(to_uyvy(v) if self.is422 else v)
            else:
                self.variants.append(self.frame)  # (cloned per lane below: other addresses, same bytes)
        p = lib.default_parameters()
        p.quality, p.restart_interval, p.verbose, p.perf_stats = self.quality, G.RESTART_AUTO, -1, 1
        if internal_rgb:
            p.color_space_internal = 1
        pi = lib.default_image_parameters()
        pi.width, pi.height = self.width, self.height
        if self.is422:
            pi.pixel_format, pi.color_space = G.P1020_422, G.YCBCR_JPEG
            p.interleaved = 1
            lib.L.gpujpeg_parameters_chroma_subsampling(C.byref(p), G.SUBSAMPLING_422)
        self.p, self.pi = p, pi
        self.p_quiet = type(p).from_buffer_copy(p)  # the timed regions run without the per-kernel events (perf_stats = 0)
        self.p_quiet.perf_stats = 0
        self.pixels = self.width * self.height
        self.raw_bytes = self.pixels * (2 if self.is422 else 3)

    def describe(self):
        if self.is422:
            return f"{self.width}x{self.height} YCbCr 4:2:2 (UYVY) q{self.quality} interleaved, restart auto"
        return f"{self.width}x{self.height} RGB 4:4:4 q{self.quality} non-interleaved, restart auto"


_TUNE = {"no_tokens": False}
_PIN = {"cpus": None}  # cores of this rank's launch threads (gpujpeg_amd.sharding.plan_affinity), None = leave the scheduler alone


def plan_pinning(local_rank, local_world, threads, ndev):
    """The launch threads of this rank go to cores of the NUMA node of its GPU, disjoint from the other ranks of the node."""
    from gpujpeg_amd.sharding import busy_cpus, gpu_local_cpus, plan_affinity
    near = []
    for r in range(local_world):
        pr = torch.cuda.get_device_properties(r % ndev)
        try:
            near.append(gpu_local_cpus("%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)))
        except AttributeError:
            near.append(None)
    _PIN["cpus"] = plan_affinity(local_rank, local_world, threads, sorted(os.sched_getaffinity(0)), near, avoid=busy_cpus())
    return {"launch_thread_cpus": _PIN["cpus"], "gpu_numa_cpus_known": near[local_rank] is not None,
            "note": "each launch thread (one per pipeline) is bound to one idle core of the NUMA node of its GPU; ranks of one node take disjoint cores"}


def pin_worker(idx):
    if _PIN["cpus"]:
        from gpujpeg_amd.sharding import pin_current_thread
        pin_current_thread(_PIN["cpus"][idx % len(_PIN["cpus"])])


_STREAMS = {}


class BenchLane(C.Structure):
    """struct gj_bench_lane of tools/bench_loop.c"""
    _fields_ = [("enc", C.c_void_p), ("dec", C.c_void_p), ("param", C.c_void_p), ("param_image", C.c_void_p), ("images", C.POINTER(C.c_void_p)),
                ("image_count", C.c_int), ("images_on_device", C.c_int), ("out", C.c_void_p), ("out_on_device", C.c_int),
                ("outs", C.POINTER(C.c_void_p)), ("out_count", C.c_int)]


_CLOOP = {"lib": None, "tried": False}
C_LOOP_OK = {"ok": True}  # (--lib / --python-loop: another build of the library is loaded, or the caller wants the interpreter's loop measured)


def c_loop(enabled=True):
    """tools/bench_loop.c: the frame loop of a launch thread in C, public API only (None: the Python loop is used)"""
    if not enabled:
        return None
    if not _CLOOP["tried"]:
        _CLOOP["tried"] = True
        path = os.path.join(ROOT, "gpujpeg_amd", "lib", "libgj_benchloop.so")
        if os.path.exists(path):
            L = C.CDLL(path)
            L.gj_bench_run.restype = C.c_int
            L.gj_bench_run.argtypes = [C.POINTER(BenchLane), C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_double), C.POINTER(C.c_size_t)]
            _CLOOP["lib"] = L
    return _CLOOP["lib"]


def run_frames_c(loop, ln, p, pi, images, on_device, out_ptr, frames, mode, jp, js, outs=None):
    """`frames` frames of one pipeline through tools/bench_loop.c; returns (jpeg pointer, size, encoder seconds, decoder seconds, bytes)"""
    arr = (C.c_void_p * len(images))(*images)
    oarr = (C.c_void_p * len(outs))(*outs) if outs else None
    lane = BenchLane(ln["enc"].h, ln["dec"].h, C.addressof(p), C.addressof(pi), arr, len(images), int(on_device), out_ptr, int(on_device),
                     oarr, len(outs) if outs else 0)
    jpeg = C.cast(jp, C.POINTER(C.c_uint8))
    size, secs, nbytes = C.c_size_t(int(js)), (C.c_double * 2)(), C.c_size_t(0)
    rc = loop.gj_bench_run(C.byref(lane), frames, {"both": 0, "encode": 1, "decode": 2}[mode], C.byref(jpeg), C.byref(size), secs, C.byref(nbytes))
    assert rc == 0, f"API call failed in the C frame loop ({rc})"
    return jpeg, size.value, secs[0], secs[1], nbytes.value


def settle_interpreter():
    """Round 3 saw every pipeline stall for 35-70 ms once, ~1000 frames into a process, and blamed the HIP runtime. It is this interpreter:
    the ctypes calls of the launch threads allocate a dozen collector-tracked objects per frame, the ~14 000th allocation starts CPython's
    first generation-2 collection, and with `import torch` alive that pass walks 170 000 objects holding the GIL (profiles/r4_07_halt_root_cause.txt:
    33-45 ms, no HIP call in flight in any thread, gone with gc.freeze()). So: collect once, then move everything imported so far to the
    permanent generation. The collector stays on; its passes now see only what the benchmark itself allocates."""
    import gc
    gc.collect()
    gc.freeze()



def lane_stream(device, index):
    """one HIP stream per pipeline index, made once per process and reused by every workload (a caller of the library keeps its streams too)"""
    key = (str(device), index)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device)
    return _STREAMS[key]


class Lanes:
    """S independent pipelines over one Spec: stream + encoder + decoder + frame + output buffer each."""

    def __init__(self, lib, spec, device, streams, host_io=False, keep_coefs=False):
        self.lib, self.spec, self.device, self.host_io = lib, spec, device, host_io
        self.p = spec.p
        self.c_loop = True
        self.lanes = []
        for si in range(max(1, streams)):
            ts = lane_stream(device, si)
            e, d = G.Encoder(lib, ts.cuda_stream), G.Decoder(lib, ts.cuda_stream)
            ln = {"stream": ts, "enc": e, "dec": d}
            if host_io:  # what a drop-in caller of the reference API has: host memory on both sides (pinned, like gpujpeg_image_load_from_file's)
                ln["frames"] = [v.cpu().pin_memory() for v in spec.variants]
                ln["outs"] = [torch.empty_like(ln["frames"][0]).pin_memory() for _ in spec.variants]
                assert e.set_option("enc_opt_out", "enc_out_val_pinned") == 0
            else:
                ln["frames"] = [v if (si == 0 and (k == 0 or v is not spec.frame)) else v.clone() for k, v in enumerate(spec.variants)]
                ln["outs"] = [torch.empty_like(spec.frame) for _ in spec.variants]
                assert e.set_option("enc_opt_out", "enc_out_val_device") == 0
            ln["frame"], ln["out"], ln["turn"] = ln["frames"][0], ln["outs"][0], 0
            if keep_coefs:
                d.keep_coefficients()
            d.init(spec.p, lib.default_image_parameters())  # turns perf_stats on for the decoder (same API as the reference)
            if spec.is422:
                d.set_output_format(G.YCBCR_JPEG, G.P1020_422)
            self.lanes.append(ln)
        torch.cuda.synchronize()

    def encode(self, ln):
        sp = self.spec
        ln["turn"] = (ln["turn"] + 1) % len(ln["frames"])  # (the Python loop rotates like tools/bench_loop.c; decode() writes the output of the same turn)
        ln["frame"], ln["out"] = ln["frames"][ln["turn"]], ln["outs"][ln["turn"]]
        if self.host_io:
            inp = G.EncoderInput()
            inp.type, inp.image = G.ENCODER_INPUT_IMAGE, ln["frame"].data_ptr()
            out, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
            assert self.lib.L.gpujpeg_encoder_encode(ln["enc"].h, C.byref(self.p), C.byref(sp.pi), C.byref(inp), C.byref(out), C.byref(size)) == 0
            return out, size.value
        return ln["enc"].encode_noclone(self.p, sp.pi, ln["frame"].data_ptr(), gpu=True)

    def set_stats(self, on):
        """perf_stats of every coder (the encoder reads it from the parameters of each call, the decoder from gpujpeg_decoder_init; neither
        reconfigures for it, src/gpujpeg_common.c:632-637)"""
        self.p = self.spec.p if on else self.spec.p_quiet
        for ln in self.lanes:
            ln["dec"].init(self.p, self.lib.default_image_parameters())
        self.warm(1)

    def decode(self, ln, jp, js):
        o = G.DecoderOutput()
        o.type = G.DECODER_OUTPUT_CUSTOM_BUFFER if self.host_io else G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER
        o.data = ln["out"].data_ptr()
        assert self.lib.L.gpujpeg_decoder_decode(ln["dec"].h, C.cast(jp, C.c_void_p), js, C.byref(o)) == 0

    def warm(self, n):
        for _ in range(max(1, n)):
            for ln in self.lanes:
                jp, js = self.encode(ln)
                self.decode(ln, jp, js)
                ln["last"] = (jp, js)
        torch.cuda.synchronize()

    def solo_kernel_ms(self, iters=10):
        """per-kernel hipEvent durations (events on the coder's stream) with one pipeline and the GPU otherwise idle"""
        ln = self.lanes[0]
        acc = np.zeros(9)
        for _ in range(iters):
            jp, js = self.encode(ln)
            torch.cuda.synchronize()
            self.decode(ln, jp, js)
            torch.cuda.synchronize()
            acc += np.array(list(ln["enc"].kernel_times()) + list(ln["dec"].kernel_times()))
        return acc / iters

    def run(self, mode, steps, reps, barrier, local_rank):
        """`steps` steps of `reps` frames per pipeline; returns (elapsed seconds, encode wall of lane 0, decode wall of lane 0,
        mean contended kernel ms of lane 0)"""
        go = threading.Event()
        walls = [0.0, 0.0]
        kms = np.zeros(9)

        stats = bool(self.p.perf_stats)
        loop = c_loop(self.c_loop)

        def worker(idx):
            torch.cuda.set_device(local_rank)  # the HIP device is per host thread
            pin_worker(idx)
            ln = self.lanes[idx]
            go.wait()
            jp, js = ln["last"]
            if loop is not None and not stats:  # the whole region in one foreign call: nothing of the interpreter between two API calls
                jp, js, te, td, nb = run_frames_c(loop, ln, self.p, self.spec.pi, [f.data_ptr() for f in ln["frames"]], not self.host_io, ln["out"].data_ptr(),
                                                  steps * reps, mode, jp, js, outs=[o.data_ptr() for o in ln["outs"]])
                if idx == 0:
                    walls[0], walls[1] = te, td
                ln["last"] = (jp, js)
                ln["mean_jpeg"] = nb / max(1, steps * reps)
                return
            for _ in range(steps * reps):
                a = time.perf_counter()
                if mode != "decode":
                    jp, js = self.encode(ln)
                b = time.perf_counter()
                if mode != "encode":
                    self.decode(ln, jp, js)
                c = time.perf_counter()
                if idx == 0:
                    walls[0] += b - a
                    walls[1] += c - b
                    if not stats:
                        continue
                    if mode != "decode":
                        kms[:5] += np.array(ln["enc"].kernel_times())
                    if mode != "encode":
                        kms[5:] += np.array(ln["dec"].kernel_times())
            ln["last"] = (jp, js)

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(self.lanes))]
        for t in threads:
            t.start()
        barrier()
        t0 = time.perf_counter()
        go.set()
        for t in threads:
            t.join()
        barrier()
        return time.perf_counter() - t0, walls[0], walls[1], kms / max(1, steps * reps)

    def close(self):
        for ln in self.lanes:
            ln["enc"].close()
            ln["dec"].close()
        self.lanes = []


def owns_direction(name):
    """Does this kernel alone move its direction's algorithmic bytes (raw pixels on one side, the entropy-coded stream on the other)? The fused
    encoders do (pixels in, tile streams out); every decoder kernel and the tail kernels of the encoder are stages."""
    return name.startswith("enc:k_encode_")


def kernel_names(spec, enc_ms, token_mode):
    whole = enc_ms[1] < 0.02 * max(1.0, spec.pixels / 33e6)  # fully fused encoder: pixels -> segment streams in one kernel (slots 0/1 empty)
    fmt = "uyvy422" if spec.is422 else "rgb444"
    # (round 4: the k_encode_* kernels leave the finished stream themselves or, for frames of more tiles than the device holds at once,
    # through k_gather; k_scan_segments + k_assemble follow k_huffman only)
    return ["enc:k_preprocess", f"enc:k_fused_{fmt}", f"enc:k_encode_{fmt}" if whole else "enc:k_huffman", "enc:k_gather" if whole else "enc:k_scan_segments", "enc:k_assemble",
            ("dec:k_huffman_decode_win" if token_mode else "dec:k_huffman_decode_seq") if spec.is422 and spec.pixels > 3e7 else ("dec:k_huffman_decode_tok" if token_mode else "dec:k_huffman_decode_par"),
            f"dec:k_idct_{'tok' if token_mode else 'fused'}_{fmt}", "dec:k_postprocess", "dec:k_markers"]


def measure(lib, spec, device, local_rank, barrier, mode="both", streams=4, steps=20, warmup=3, min_seconds=0.5, host_io=False, keep_coefs=False,
            want_solo=False):
    """Run one workload; returns a dict with throughput and timings."""
    L = Lanes(lib, spec, device, streams, host_io=host_io, keep_coefs=keep_coefs)
    L.c_loop = C_LOOP_OK["ok"]
    sync = lambda: torch.cuda.synchronize()
    L.warm(2)  # buffers allocated, tables uploaded, a stream for the decoders to start from
    solo = L.solo_kernel_ms() if want_solo else None
    if not os.environ.get("BENCH_TIMED_STATS"):  # (developer A/B: keep the per-kernel events in the timed regions as round 3 did)
        L.set_stats(False)  # nothing below records per-kernel events until the short contended-kernel region at the end
    # fix the batch: frames per pipeline and step so that `steps` steps last >= min_seconds
    L.run(mode, 1, 8, sync, local_rank)  # (the first frames of a new coder are slower than the steady state)
    t_probe, *_ = L.run(mode, 1, 8, sync, local_rank)
    per_frame = max(t_probe / 8, 1e-6)
    reps = max(1, int(np.ceil(1.15 * min_seconds / (steps * per_frame)))) if min_seconds > 0 else 1
    if warmup > 0:  # W untimed steps of exactly the shape of the timed ones (same threads, same frames per step)
        L.run(mode, warmup, reps, sync, local_rank)
    elapsed, enc_wall, dec_wall, _ = L.run(mode, steps, reps, barrier, local_rank)
    kms = np.zeros(9)
    if want_solo:  # per-kernel durations with all pipelines running: their own short region, with the events on
        L.set_stats(True)
        *_, kms = L.run(mode, 1, max(8, reps // 2), sync, local_rank)
    jsize = int(round(L.lanes[0].get("mean_jpeg") or L.lanes[0]["last"][1]))  # (mean over the frames of the rotation)
    S = len(L.lanes)
    L.close()
    return {"elapsed": elapsed, "reps": reps, "streams": S, "enc_wall": enc_wall, "dec_wall": dec_wall, "kernel_ms": kms, "solo_ms": solo,
            "jpeg_bytes": jsize, "frames": S * steps * reps}


def cpu_baseline(spec, frame_host, frames=2):
    """the reference's CPU path on ONE host thread: its own host C (driver, writer/reader, CPU Huffman coders)
    + the restated scalar stages for colour / fDCT+quantiser / dequantiser+IDCT (oracle/_ref/libgpujpeg_refcpu.so, gcc -O2)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    path = os.path.join(O.HERE, "_ref", "libgpujpeg_refcpu.so")
    w, h, q, is422 = spec.width, spec.height, spec.quality, spec.is422
    t0 = time.time()
    if os.path.exists(path):
        kind = "reference"
        ref = G.Library(path)
        enc, dec = G.Encoder(ref), G.Decoder(ref)
        p = ref.default_parameters()
        p.restart_interval, p.verbose, p.quality = G.RESTART_AUTO, -1, q
        pi = ref.default_image_parameters()
        pi.width, pi.height = w, h
        if is422:
            pi.pixel_format, pi.color_space = G.P1020_422, G.YCBCR_JPEG
            p.interleaved = 1
            dec.set_output_format(G.YCBCR_JPEG, G.P1020_422)
        t_enc = t_dec = 0.0
        for _ in range(frames):
            a = time.time()
            jpeg = enc.encode(p, pi, frame_host)
            b = time.time()
            dec.decode(jpeg)
            t_enc += b - a
            t_dec += time.time() - b
        ref.L.gjref_time_idct_cpu.restype = C.c_double
        ref.L.gjref_time_idct_cpu.argtypes = [C.c_void_p]
        idct_cpu = float(ref.L.gjref_time_idct_cpu(dec.h))
    else:
        kind, idct_cpu = "port", None
        img = (O.make_image(w, h, pixel_format=3, color_space=3, quality=q, interleaved=1) if is422 else O.make_image(w, h, quality=q))
        t_enc = t_dec = 0.0
        for _ in range(frames):
            a = time.time()
            jpeg = O.encode(img, frame_host)
            b = time.time()
            O.decode(jpeg, 3, 3) if is422 else O.decode(jpeg)
            t_enc += b - a
            t_dec += time.time() - b
    dt = time.time() - t0
    px = w * h * frames
    return {"value": round(px / (t_enc + t_dec) / 1e6, 3), "unit": "Mpix/s", "cores": 1, "kind": kind,
            "encode_mpix_s": round(px / t_enc / 1e6, 3), "decode_mpix_s": round(px / t_dec / 1e6, 3),
            "idct_cpu_s": None if idct_cpu is None else round(idct_cpu, 3),
            "sample": f"{frames} x encode+decode of the {spec.describe()} frame on one thread ({dt:.1f} s): reference host C (writer / reader / CPU Huffman, "
                      f"gcc -O2) + restated scalar colour, fDCT+quantiser, dequantiser+IDCT; idct_cpu_s = the reference's own integer gpujpeg_idct_cpu "
                      f"(src/gpujpeg_dct_cpu.c:178-203) on that frame's coefficients"}


_CPU_CHILD = r"""
import sys, time
import numpy as np
root, lib, path, w, h, q, is422, frames = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7] == "1", int(sys.argv[8])
sys.path.insert(0, root)
from gpujpeg_amd import libgpujpeg as G
frame = np.load(path)
ref = G.Library(lib)
enc, dec = G.Encoder(ref), G.Decoder(ref)
p = ref.default_parameters(); p.restart_interval, p.verbose, p.quality = G.RESTART_AUTO, -1, q
pi = ref.default_image_parameters(); pi.width, pi.height = w, h
if is422:
    pi.pixel_format, pi.color_space = G.P1020_422, G.YCBCR_JPEG; p.interleaved = 1; dec.set_output_format(G.YCBCR_JPEG, G.P1020_422)
t0 = time.time()
for _ in range(frames):
    dec.decode(enc.encode(p, pi, frame))
print(time.time() - t0)
"""


def cpu_baseline_all_cores(spec, frame_host, max_procs=64):
    """the reference's CPU code admits frame-level parallelism only -- one independent encode+decode per process
    on the host's cores (capped), the same frame in each; aggregate Mpix/s over the slowest process. -O3 -march=native build."""
    import subprocess
    import tempfile
    lib = os.path.join(ROOT, "oracle", "_ref", "libgpujpeg_refcpu_native.so")
    if not os.path.exists(lib):
        return None
    procs = max(1, min(max_procs, (os.cpu_count() or 1)))
    with tempfile.NamedTemporaryFile(suffix=".npy", dir="/dev/shm" if os.path.isdir("/dev/shm") else None, delete=False) as f:
        np.save(f, frame_host)
        path = f.name
    try:
        t0 = time.time()
        kids = [subprocess.Popen([sys.executable, "-c", _CPU_CHILD, ROOT, lib, path, str(spec.width), str(spec.height), str(spec.quality),
                                  "1" if spec.is422 else "0", "1"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(procs)]
        inner = [float(k.communicate(timeout=300)[0].decode().strip().splitlines()[-1]) for k in kids]
        wall = time.time() - t0
    finally:
        os.unlink(path)
    return {"value": round(spec.pixels * procs / max(inner) / 1e6, 3), "unit": "Mpix/s", "cores": procs, "host_cores": os.cpu_count(), "kind": "reference",
            "sample": f"{procs} processes x 1 encode+decode of the same {spec.width}x{spec.height} frame at once (frame-level parallelism, the only kind the "
                      f"reference's CPU code admits), gcc -O3 -march=native; slowest process {max(inner):.2f} s, {wall:.1f} s including start-up",
            "cap": "64 processes is this host's best, not a shortcut: the work is bound by memory bandwidth (64 processes = 13 x one), and with 128 / "
                   "256 processes (one per core) the same box measured 164.6 / 158.4 Mpix/s in 27 / 58 s (round 3, DESIGN 4.2)"}


def run_batch(args, lib, device, local_rank, rank, world, width, height, emit=True):
    """BASELINE.json config 5 / SURVEY.md 8(d): a batch of independent frames, frame i seeded 12345 + i, sharded over the
    ranks by gpujpeg_amd.sharding.shard_frames (static round-robin, no data-path collective). Every rank keeps its shard
    resident in HBM, S pipelines (stream + encoder + decoder + host thread) walk disjoint slices of it."""
    import torch.distributed as dist
    from gpujpeg_amd.sharding import barrier_and_max, gather_counts, shard_frames
    mine = shard_frames(args.batch, rank, world)
    red = device if (world == 1 or dist.get_backend() == "nccl") else None  # where the reductions of the timing live (gloo: host)
    host_io = getattr(args, "batch_io", "device") == "host"
    is422 = args.workload.endswith("422")  # packed YCbCr 4:2:2 (UYVY), interleaved scan, q90 when the quality was left at 75 (like Spec)
    quality = 90 if (is422 and args.quality == 75) else args.quality
    frames = [synth_frame(lib, width, height, args.pattern, 12345 + i, device) for i in mine]
    if is422:
        frames = [to_uyvy(f) for f in frames]
    if host_io:  # pinned host memory on both sides: the caller of the reference API (gpujpeg_image_load_from_file returns pinned memory too)
        frames = [f.cpu().pin_memory() for f in frames]
        torch.cuda.empty_cache()
    # --batch-api batch: the frames of a pipeline go through gpujpeg_amd_encoder_encode_batch / gpujpeg_amd_decoder_decode_batch (every kernel
    # launched once per chunk of frames, the frame is a grid dimension) instead of one libgpujpeg call per frame
    batch_api = getattr(args, "batch_api", "frame") == "batch"
    direction = getattr(args, "mode", "both") if batch_api else "both"  # (one direction alone: the batch calls only)
    S = max(1, min(getattr(args, "batch_streams", 0) or args.streams, len(frames))) if batch_api else max(1, min(args.streams, len(frames)))
    p = lib.default_parameters()
    p.quality, p.restart_interval, p.verbose = quality, G.RESTART_AUTO, -1
    pi = lib.default_image_parameters()
    pi.width, pi.height = width, height
    if is422:
        pi.pixel_format, pi.color_space = G.P1020_422, G.YCBCR_JPEG
        p.interleaved = 1
        lib.L.gpujpeg_parameters_chroma_subsampling(C.byref(p), G.SUBSAMPLING_422)
    lanes = []
    for si in range(S):
        ts = lane_stream(device, si)
        e, d = G.Encoder(lib, ts.cuda_stream), G.Decoder(lib, ts.cuda_stream)
        if is422:
            d.set_output_format(G.YCBCR_JPEG, G.P1020_422)
        assert e.set_option("enc_opt_out", "enc_out_val_pinned" if host_io else "enc_out_val_device") == 0
        ln = {"frames": frames[si::S], "out": torch.empty_like(frames[0]).pin_memory() if host_io else torch.empty_like(frames[0]),
              "enc": e, "dec": d, "bytes": 0, "digest": [], "batched": None}
        if batch_api:  # the pipeline's frames back to back, and room for as many decoded frames
            ln["stack"] = torch.stack(ln["frames"]).contiguous()
            if host_io:
                ln["stack"] = ln["stack"].pin_memory()
            ln["frames"] = [ln["stack"][k] for k in range(ln["stack"].shape[0])]
            ln["out_stack"] = torch.empty_like(ln["stack"]).pin_memory() if host_io else torch.empty_like(ln["stack"])
            ln["out"] = ln["out_stack"][-1]
        lanes.append(ln)
    if batch_api:
        del frames
        torch.cuda.empty_cache()
    torch.cuda.synchronize()

    loop = c_loop(C_LOOP_OK["ok"])

    def one_pass(ln, digest=False):
        if batch_api:
            n = len(ln["frames"])
            raw = ln["stack"][0].numel()
            # --mode encode / decode: one direction per pass (decode: the streams of the warm-up's encode stay in the encoder's buffer)
            if direction != "decode" or "streams" not in ln:
                ptrs, sizes = ln["enc"].encode_batch_noclone(p, pi, ln["stack"].data_ptr(), n, stride=raw, gpu=True)
                ln["streams"] = (ptrs, sizes)
            ptrs, sizes = ln["streams"]
            if direction != "encode":
                stride = ptrs[1] - ptrs[0] if n > 1 else (sizes[0] + 64 + 15) & ~15
                ln["dec"].decode_batch(None, device_out=ln["out_stack"].data_ptr(), out_stride=raw, device_in=ptrs[0], in_stride=stride, sizes=sizes)
            ln["bytes"] = sum(sizes)
            ln["batched"] = (ln["enc"].last_batch(), ln["dec"].last_batch())
            return
        if loop is not None:  # the frames of this pipeline, in order, through tools/bench_loop.c
            *_, ln["bytes"] = run_frames_c(loop, ln, p, pi, [f.data_ptr() for f in ln["frames"]], not host_io, ln["out"].data_ptr(),
                                           len(ln["frames"]), "both", None, 0)
            return
        nbytes = 0
        for f in ln["frames"]:
            if host_io:
                inp = G.EncoderInput()
                inp.type, inp.image = G.ENCODER_INPUT_IMAGE, f.data_ptr()
                jp, size = C.POINTER(C.c_uint8)(), C.c_size_t(0)
                assert lib.L.gpujpeg_encoder_encode(ln["enc"].h, C.byref(p), C.byref(pi), C.byref(inp), C.byref(jp), C.byref(size)) == 0
                js = size.value
            else:
                jp, js = ln["enc"].encode_noclone(p, pi, f.data_ptr(), gpu=True)
            o = G.DecoderOutput()
            o.type, o.data = G.DECODER_OUTPUT_CUSTOM_BUFFER if host_io else G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, ln["out"].data_ptr()
            assert lib.L.gpujpeg_decoder_decode(ln["dec"].h, C.cast(jp, C.c_void_p), js, C.byref(o)) == 0
            nbytes += js
        ln["bytes"] = nbytes

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    go = threading.Event()

    pass_ms = [[] for _ in range(S)]

    def worker(idx, passes, wait):
        torch.cuda.set_device(local_rank)
        pin_worker(idx)
        if wait:
            go.wait()
        for _ in range(passes):
            t_ = time.perf_counter()
            one_pass(lanes[idx])
            pass_ms[idx].append(round((time.perf_counter() - t_) * 1e3, 2))

    elapsed = 0.0
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
    verified = None
    if args.verify:  # every frame of this rank's shard against the CPU oracle: encoder bytes (sha256) and decoded samples
        verified = verify_batch(lib, lanes, p, pi, width, height, quality, mine, S, is422)
    elapsed = barrier_and_max(elapsed, red)
    total = gather_counts(len(mine), red)
    jpeg_bytes = gather_counts(sum(ln["bytes"] for ln in lanes), red)
    if verified is not None:
        verified = gather_counts(0 if verified else 1, red) == 0
    for ln in lanes:
        ln["enc"].close()
        ln["dec"].close()
    fps = total * args.steps / elapsed
    result = {
        "metric": f"frames/s {'encode+decode' if direction == 'both' else direction + ' only'} (batch of {args.batch} {args.workload} frames)", "value": round(fps, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "dtype_detail": DTYPE_DETAIL,
        "data": f"synthetic ({args.pattern}), {args.batch} distinct {width}x{height} frames (seed 12345 + i) "
                + ("in pinned host memory, results to pinned host memory (PCIe both ways)" if host_io else "resident in HBM") + ", sharded round-robin",
        "config": {"workload": f"{args.batch} x {width}x{height} " + (f"YCbCr 4:2:2 (UYVY) q{quality} interleaved" if is422 else f"RGB 4:4:4 q{quality} non-interleaved") + ", restart auto, encode then decode of "
                               f"every frame per step", "frames_total": total, "frames_per_gpu": len(mine), "streams_per_gpu": S,
                   "jpeg_bytes_total": jpeg_bytes, "parallelism": f"frame-sharded x{world}, no collective", "io": "host" if host_io else "device",
                   "api": ("gpujpeg_amd_encoder_encode_batch + gpujpeg_amd_decoder_decode_batch: one set of launches per chunk of frames; (frames coded by "
                           f"batched launches, one by one) encoder / decoder of pipeline 0's last pass: {lanes[0]['batched']}") if batch_api
                          else "gpujpeg_encoder_encode + gpujpeg_decoder_decode per frame",
                   "cpu_affinity_rank0": getattr(args, "affinity", None),
                   "pass_ms_per_pipeline": [pm[-args.steps:] for pm in pass_ms],
                   "hbm_free_gb_at_end": round(torch.cuda.mem_get_info(device)[0] / 2**30, 1), "hbm_total_gb": round(torch.cuda.mem_get_info(device)[1] / 2**30, 1)},
        "mpix_s": round(fps * width * height / 1e6, 2), "psnr_last_frame_db": round(10 * np.log10(255.0 ** 2 / max(mse, 1e-9)), 2)}
    if verified is not None:
        result["verified_bit_exact"] = verified
    if emit:
        if rank == 0:
            print(json.dumps(result), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    return result


def verify_batch(lib, lanes, p, pi, width, height, quality, mine, S, is422=False):
    """oracle check of a rank's shard (tests / --verify only; the oracle is the checker, never the thing measured)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    ok = True
    img = O.make_image(width, height, pixel_format=3, color_space=3, quality=quality, interleaved=1) if is422 else O.make_image(width, height, quality=quality)
    odec = (lambda j: O.decode(j, 3, 3)[0]) if is422 else (lambda j: O.decode(j)[0])
    hip = C.cdll.LoadLibrary("libamdhip64.so")
    for si, ln in enumerate(lanes):
        if "stack" in ln:  # the batch calls: every stream and every decoded frame of one more pass
            n, raw = len(ln["frames"]), ln["stack"][0].numel()
            ptrs, sizes = ln["enc"].encode_batch_noclone(p, pi, ln["stack"].data_ptr(), n, stride=raw, gpu=True)
            streams = []
            for ptr, js in zip(ptrs, sizes):
                jt = np.empty(js, np.uint8)
                hip.hipMemcpy(C.c_void_p(jt.ctypes.data), C.c_void_p(ptr), C.c_size_t(js), 2)
                streams.append(jt)
            ln["out_stack"].zero_()
            ln["dec"].decode_batch(None, device_out=ln["out_stack"].data_ptr(), out_stride=raw, device_in=ptrs[0],
                                   in_stride=ptrs[1] - ptrs[0] if n > 1 else (sizes[0] + 79) & ~15, sizes=sizes)
            torch.cuda.synchronize()
            for k in range(n):
                want = O.encode(img, ln["stack"][k].cpu().numpy().reshape(-1))
                ok = ok and bool(np.array_equal(streams[k], want)) and bool(np.array_equal(ln["out_stack"][k].cpu().numpy().reshape(-1), odec(want)))
            continue
        for f in ln["frames"]:
            host = f.cpu().numpy().reshape(-1)
            want = O.encode(img, host)
            jp, js = ln["enc"].encode_noclone(p, pi, f.data_ptr(), gpu=True)
            jt = torch.empty(js, dtype=torch.uint8, device=f.device)
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(jt.data_ptr()), C.cast(jp, C.c_void_p), C.c_size_t(js), 3)
            o = G.DecoderOutput()
            o.type, o.data = G.DECODER_OUTPUT_CUSTOM_CUDA_BUFFER, ln["out"].data_ptr()
            assert lib.L.gpujpeg_decoder_decode(ln["dec"].h, C.cast(jp, C.c_void_p), js, C.byref(o)) == 0
            torch.cuda.synchronize()
            ok = ok and bool(np.array_equal(jt.cpu().numpy(), want)) and bool(np.array_equal(ln["out"].cpu().numpy().reshape(-1), odec(want)))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="8k", choices=sorted(WORKLOADS))
    ap.add_argument("--pattern", default="natural", choices=["natural", "noise", "gradient", "camera"])
    ap.add_argument("--batch-io", default="device", choices=["device", "host"],
                    help="--batch: frames and results resident in HBM (default) or in pinned host memory on both sides (what a drop-in caller has)")
    ap.add_argument("--batch-api", default="frame", choices=["frame", "batch"],
                    help="--batch: `frame` = one libgpujpeg call per frame and direction (the reference's API), `batch` = the frames of a pipeline through "
                         "gpujpeg_amd_encoder_encode_batch / gpujpeg_amd_decoder_decode_batch (MI355X extension: the frame is a grid dimension of every kernel)")
    ap.add_argument("--batch-streams", type=int, default=2, help="pipelines per GPU with --batch-api batch (a batch call fills the device by itself: two hide each other's host gaps)")
    ap.add_argument("--python-loop", action="store_true", help="drive the API calls of the timed regions from Python instead of tools/bench_loop.c")
    ap.add_argument("--pin", default="auto", choices=["auto", "on", "off"],
                    help="bind the launch threads to idle cores of the GPU's NUMA node: `auto` (default) does it when several ranks share the node -- "
                         "that is when placement matters -- and leaves a single rank to the scheduler, which steers clear of cores other tenants keep busy")
    ap.add_argument("--no-pin", action="store_true", help="same as --pin off")
    ap.add_argument("--quality", type=int, default=75)
    ap.add_argument("--batch", type=int, default=0,
                    help="BASELINE.json config 5: a fixed batch of this many distinct frames (seeds 12345 + i) sharded over the ranks "
                         "(strong scaling, frames/s); a step is one pass over the whole batch")
    ap.add_argument("--streams", type=int, default=4, help="independent encoder+decoder pairs per GPU, each on its own HIP stream and host thread "
                    "(hides the host side of one call behind the kernels of the others)")
    ap.add_argument("--mode", default="both", choices=["both", "encode", "decode"],
                    help="what a timed step does: encode then decode (the headline metric), or only one direction")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="lower bound of the timed region: the frames per pipeline and step are chosen for it")
    ap.add_argument("--lean", action="store_true", help="headline + roofline only (no other workloads, no full-API run, no CPU baselines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true")
    ap.add_argument("--internal-rgb", action="store_true", help="code RGB without colour transform (tuning aid; not the headline config)")
    ap.add_argument("--calibrate", action="store_true", help="run a 256 MiB device fill + copy first (known byte counts for calibrating PMC traffic counters)")
    ap.add_argument("--keep-coefs", action="store_true", help="decoder keeps its coefficients in HBM (adds the per-frame clear; tuning aid)")
    ap.add_argument("--lib", default=None, help="another build of the library (A/B runs of compile-time variants: make -C gpujpeg_amd/csrc variant NAME=...)")
    ap.add_argument("--host-io", action="store_true", help="the timed region with pinned host buffers in and out (what `full_api` reports; tuning aid, not the headline)")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME[=VALUE]", help="developer setting for every coder of this run (gpujpeg_amd_tuning; INTEGRATION.md), repeatable")
    ap.add_argument("--verify", action="store_true", help="check the results of the last frame(s) against the oracle (slow)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as ONE process (the driver's command shape): become the launcher of N ranks, one per GPU,
        # under torch.distributed.run on this node; the ranks re-enter main() with RANK / LOCAL_RANK / WORLD_SIZE set.
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(1, ndev)  # (more ranks than devices: the 2-process proof on one GPU, tests/test_gpu_multiproc.py)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if ndev >= world else "gloo", **({"device_id": device} if ndev >= world else {}))

    lib = G.Library(args.lib)  # raises if the HIP library has not been built: there is no fallback
    # developer settings (forced kernel paths, A/B): the library does not read the environment; this tool hands over what its environment and --tune name
    G.apply_environment_settings(lib)
    for t in args.tune:
        assert lib.tuning(t), f"unknown developer setting {t}"
    _TUNE["no_tokens"] = bool(os.environ.get("GJ_DEC_NO_TOKENS")) or any(t.split("=")[0] == "GJ_DEC_NO_TOKENS" for t in args.tune)
    assert lib.L.gpujpeg_init_device(dev_index, 0) == 0
    settle_interpreter()
    C_LOOP_OK["ok"] = not (args.lib or args.python_loop)
    width, height = WORKLOADS[args.workload]
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    pin = "off" if args.no_pin else args.pin
    args.affinity = plan_pinning(local_rank, local_world, max(1, args.streams), max(1, ndev)) if pin == "on" or (pin == "auto" and local_world > 1) else None
    if args.batch:
        return run_batch(args, lib, device, dev_index, rank, world, width, height)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.calibrate:
        cal_a = torch.zeros(2 ** 26, dtype=torch.int32, device=device)
        cal_b = cal_a.clone()
        torch.cuda.synchronize()
        del cal_a, cal_b

    spec = Spec(lib, args.workload, args.pattern, args.quality, device, 12345 + rank, internal_rgb=args.internal_rgb)
    head = measure(lib, spec, device, dev_index, barrier, mode=args.mode, streams=args.streams, steps=args.steps, warmup=args.warmup,
                   min_seconds=args.min_seconds, keep_coefs=args.keep_coefs, want_solo=True, host_io=args.host_io)
    # whole-job figures: max over ranks of the elapsed time, SUM over ranks of the frames each rank really coded (every rank sizes its
    # own batch from its own probe, so the counts differ by a few per cent)
    from gpujpeg_amd.sharding import barrier_and_max, gather_counts
    red = device if (world == 1 or ndev >= world) else None
    elapsed = barrier_and_max(head["elapsed"], red)
    frames_all = gather_counts(head["frames"], red)
    ranks_seen = gather_counts(1, red)

    if rank == 0:
        S, reps, jsize = head["streams"], head["reps"], head["jpeg_bytes"]
        nblocks = ((width + 7) // 8) * ((height + 7) // 8) * (2 if spec.is422 else 3)
        token_mode = nblocks >= (900000 if spec.is422 else 300000) and jsize <= (12 if spec.is422 else 16) * nblocks and not args.keep_coefs and not _TUNE["no_tokens"]
        names = kernel_names(spec, head["solo_ms"], token_mode)
        alg = spec.raw_bytes + jsize  # encoder: raw in + JPEG out; decoder: JPEG in + raw out (the same sum)
        solo, cont = head["solo_ms"], head["kernel_ms"]
        standard = spec.quality == (90 if spec.is422 else 75) and args.pattern == "natural" and not args.internal_rgb
        traffic = load_traffic("kernels", args.workload) if standard else {}

        def roof(name, ms):
            ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "ms": round(float(ms), 4), "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(alg)}

        def stage(name, ms):
            #  a kernel that is ONE STAGE of its direction -- entropy decoder, IDCT, k_gather, the marker scan -- does not move the
            # direction's algorithmic bytes: it is reported with the bytes it touched (PMC passes under profiles/, when taken on these sources)
            # against the peak and with its vector-issue fraction, not with an HBM `frac` of bytes it never sees
            if owns_direction(name):
                return dict(roof(name, ms), owns_direction_bytes=True, traffic=traffic.get(name), **issue(name, ms))
            t = traffic.get(name)
            touched = None if not t or ms <= 0 else round(t / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            return dict({"kernel": name, "ms": round(float(ms), 4), "owns_direction_bytes": False, "bytes_touched": t, "bytes_touched_frac_of_peak": touched}, **issue(name, ms))

        live = [i for i in range(9) if solo[i] > 0.006]  # (event slots of kernels this configuration does not launch hold only the gap between two events)
        dom = max(live, key=lambda i: solo[i])
        # what an event pair measures with NOTHING between the two records (the slots of kernels this configuration does not launch): the part of every
        # `ms` below that is the event records themselves -- rocprofv3's kernel durations (profiles/) correspond to ms - event_gap_ms
        idle_slots = [float(solo[i]) for i in range(9) if 0.0005 < solo[i] <= 0.006]
        event_gap = round(sum(idle_slots) / len(idle_slots), 4) if idle_slots else None
        valu = load_traffic("valu_insts", args.workload) if traffic else {}

        cyc, share4 = load_isa_mix()

        def issue(name, ms):
            n = valu.get(name)
            if not n or ms <= 0:
                return {}
            s4 = share4.get(name, 1.0)  # (a kernel without a counted mix: the slow class for all, the conservative floor)
            per = lambda c: n * c / SIMDS / CLOCK_HZ * 1e3  # noqa: E731
            weighted = per((1.0 - s4) * cyc["valu2"] + s4 * cyc["valu4"])
            return {"valu_insts": n, "valu_share_4cycle_class": s4, "valu_issue_floor_ms": round(weighted, 4),
                    "valu_issue_floor_ms_if_all_2cycle": round(per(cyc["valu2"]), 4), "valu_issue_floor_ms_if_all_4cycle": round(per(cyc["valu4"]), 4),
                    "valu_issue_frac": round(weighted / ms, 3)}

        by_kernel = [stage(names[i], solo[i]) for i in live]
        r = roof(names[dom], solo[dom])
        enc_total, dec_total = float(sum(solo[i] for i in live if i < 5)), float(sum(solo[i] for i in live if i >= 5))  # (slots of kernels that were not launched hold the gap between two events)
        n_enc, n_dec = sum(1 for i in live if i < 5), sum(1 for i in live if i >= 5)
        gap = event_gap or 0.0  # (an event pair brackets the kernel AND one event record: rocprofv3's kernel durations are the net values)
        enc_net, dec_net = max(enc_total - n_enc * gap, 0.0), max(dec_total - n_dec * gap, 0.0)

        def dir_frac(nbytes, ms):
            return round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else None
        result = {
            "metric": ("Mpix/s encode+decode (8K RGB q75)" if args.workload == "8k" else f"Mpix/s encode+decode ({args.workload})") if args.mode == "both"
                      else f"Mpix/s {args.mode} only ({args.workload})",
            "value": round(spec.pixels * frames_all / elapsed / 1e6, 2), "unit": "Mpix/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "frames_all_ranks": frames_all, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "dtype_detail": DTYPE_DETAIL,
            "data": f"synthetic ({args.pattern}), {S * ROTATE} {width}x{height} frames per rank resident in HBM: every pipeline walks {ROTATE} distinct frames "
                    f"and {ROTATE} output buffers round-robin (no frame is re-read out of the 256 MB Infinity Cache)",
            "config": {"workload": spec.describe() + (" (7680x4320 -> 36)" if args.workload == "8k" else "") + ", encode then decode of every frame",
                       "frames_per_step_per_gpu": S * reps, "frames_per_step_per_pipeline": reps, "streams_per_gpu": S, "timed_seconds": round(elapsed, 3),
                       "untimed_before": f"16 frames per pipeline (sizing of the step), then {args.warmup} warm-up steps of the timed shape; the interpreter's "
                                         "objects are frozen out of the garbage collector first (gc.freeze: round 3's 'stream halt' was a generation-2 collection)",
                       "jpeg_bytes": int(jsize), "parallelism": f"frame-sharded x{world}, no collective",
                       "timed_with": "perf_stats = 0 (no per-kernel events in the timed region; `roofline.contended` comes from its own short region); "
                                     + ("frame loop of every launch thread in C (tools/bench_loop.c, public API only)" if c_loop(C_LOOP_OK["ok"]) else "frame loop in Python"),
                       "cpu_affinity_rank0": args.affinity},
            # API calls of pipeline 0 alone (one call at a time per pipeline; the aggregate of all pipelines is `value`)
            "encode_mpix_s_pipeline0": round(spec.pixels * args.steps * reps / head["enc_wall"] / 1e6, 2) if args.mode != "decode" else None,
            "decode_mpix_s_pipeline0": round(spec.pixels * args.steps * reps / head["dec_wall"] / 1e6, 2) if args.mode != "encode" else None,
            "roofline": {"bound": "hbm", "kernel": r["kernel"], "achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac"],
                         "frac_basis": ("the dominant kernel's own bytes: a fused encoder moves its direction's algorithmic bytes (raw in + JPEG out) alone"
                                        if owns_direction(names[dom]) else
                                        "FLATTERING for this kernel: `frac` divides the whole decode direction's algorithmic bytes (JPEG in + raw out) by the duration "
                                        "of ONE of its stage kernels, which never writes a pixel; the figures the driver's clock and the profiles support are "
                                        "frac_path, frac_encode_direction, frac_decode_direction and frac_hbm_read_encode below"),
                         #  the four fractions a reader can recompute: whole path from the driver-timed region, each direction from the
                         # sum of its kernels' solo durations (hipEvents net of the event record, = rocprofv3's kernel durations under profiles/),
                         # and north_star's own definition: raw input bytes of the encoder / encoder direction time / peak, target 0.70
                         "frac_path": round(2 * alg * frames_all / elapsed / 1e9 / HBM_PEAK_GBS, 5) if args.mode == "both" else None,
                         "frac_path_basis": "(encode + decode algorithmic bytes per frame) / (ms_per_step / frames_per_step_per_gpu) / peak: every kernel, launch gap and "
                                            "host call of the timed region included, four pipelines sharing the device",
                         "frac_encode_direction": dir_frac(alg, enc_net), "frac_decode_direction": dir_frac(alg, dec_net),
                         "frac_hbm_read_encode": dir_frac(spec.raw_bytes, enc_net), "target": 0.70,
                         "frac_hbm_read_encode_basis": "BASELINE.json north_star: raw input bytes of the encoder (HBM read) / sum of the encoder kernels' solo durations / "
                                                       "8 TB/s; the target is 0.70. These kernels are bound by vector-instruction issue (valu_issue_frac), not by HBM: DESIGN 4",
                         "direction_ms_net_of_event_gap": {"encode": round(enc_net, 4), "decode": round(dec_net, 4), "encode_kernels": n_enc, "decode_kernels": n_dec},
                         "traffic": traffic.get(names[dom]), "ms": r["ms"], "algorithmic_bytes_per_launch": int(alg), **issue(names[dom], solo[dom]),
                         "valu_note": "valu_issue_frac: SQ_INSTS_VALU per launch (profiles/r6_traffic.json) x the cycles of the kernel's instruction class mix "
                                      "(2.4 / 4.3 cycles per wave64 instruction, profiles/r6_isa_mix.json) / 1024 SIMDs / 2.4 GHz / duration -- the roofline that "
                                      "actually bounds these kernels; the all-2-cycle and all-4-cycle floors bracket it",
                         "timing": "average hipEvent duration over 10 solo launches inside this run (one pipeline, GPU otherwise idle, events on the "
                                   "coder's stream); profiles/r6_* hold the rocprofv3 --kernel-trace --stats summary of the same configuration",
                         "dominant": ("the launch with the longest solo duration of an encoded and decoded frame. Since the encoder's last tiles are split "
                                      "(round 5: k_encode_rgb444 84 -> 75 us alone) that is the token decoder on most runs, a stage of the decode direction: "
                                      "`achieved` prices the direction's algorithmic bytes per frame (SURVEY 8(d)) against its duration, `traffic` is what it "
                                      "touches itself; by_kernel / by_direction carry the others"),
                         "event_gap_ms": event_gap,
                         "ms_net_of_event_gap": None if event_gap is None else round(r["ms"] - event_gap, 4),
                         "frac_net_of_event_gap": None if event_gap is None else round(alg / ((r["ms"] - event_gap) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "event_gap_note": "hipEvent duration of an event pair with no launch between the records, from the library's event slots of kernels this "
                                           "configuration does not launch; `ms` and `frac` above are the raw event durations (kernel + one event record), the "
                                           "rocprofv3 durations under profiles/ agree with the net values",
                         "by_kernel": by_kernel,
                         "solo_vs_filled": None,  # (filled in by extras(): the same kernel's per-frame cost when the device is full)
                         "by_direction": {"encode": dict(roof("all encoder kernels", enc_total)), "decode": dict(roof("all decoder kernels", dec_total))},
                         "contended": dict(roof(names[dom], cont[dom]), concurrent_pipelines=S,
                                           kernel_ms={names[i]: round(float(cont[i]), 4) for i in live})},
        }
        if args.verify:
            result.update(verify_headline(lib, spec, device, args))
        if world == 1 and not args.lean:
            extras(result, args, lib, spec, device, dev_index, barrier)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def verify_headline(lib, spec, device, args):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    L = Lanes(lib, spec, device, 1)
    ln = L.lanes[0]
    jp, js = L.encode(ln)
    L.decode(ln, jp, js)
    torch.cuda.synchronize()
    host = ln["frame"].cpu().numpy().reshape(-1)  # (the frame of the rotation this turn coded)
    img = (O.make_image(spec.width, spec.height, pixel_format=3, color_space=3, quality=spec.quality, interleaved=1) if spec.is422
           else O.make_image(spec.width, spec.height, quality=spec.quality, color_space_internal=1 if args.internal_rgb else 3))
    want = O.encode(img, host)
    jt = torch.empty(js, dtype=torch.uint8, device=device)
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(jt.data_ptr()), C.cast(jp, C.c_void_p), C.c_size_t(js), 3)
    torch.cuda.synchronize()
    r = {"verified_bit_exact_encode": bool(np.array_equal(jt.cpu().numpy(), want)),
         "verified_bit_exact_decode": bool(np.array_equal(ln["out"].cpu().numpy().reshape(-1), (O.decode(want, 3, 3) if spec.is422 else O.decode(want))[0]))}
    L.close()
    return r


def extras(result, args, lib, spec, device, dev_index, barrier):
    """The bounded extra runs of rank 0 at N = 1 (see the module docstring)."""
    def brief(spec_, m):
        px = spec_.pixels * m["frames"]
        return {"mpix_s": round(px / m["elapsed"] / 1e6, 1), "ms_per_frame": round(m["elapsed"] / m["frames"] * 1e3, 4), "jpeg_bytes": m["jpeg_bytes"],
                "streams": m["streams"], "timed_seconds": round(m["elapsed"], 3)}

    if args.mode == "both":
        for mode in ("encode", "decode"):
            m = measure(lib, spec, device, dev_index, barrier, mode=mode, streams=args.streams, steps=5, warmup=2, min_seconds=0.3)
            result[f"{mode}_only"] = dict(brief(spec, m), note="device resident (GPU_IMAGE in / stream in HBM, HBM buffer out): the reference's 'w/o PCIe' figure")
        # (VERDICT r4 #7b) the contract's roofline is the dominant kernel's SOLO duration, which carries a lone launch's ramp and under-filled last
        # generation of workgroups; the same kernel's cost per frame when the device is full = its share of the encode-only rate of four pipelines
        rf = result["roofline"]
        # (of the encoder's main kernel, whichever kernel is the longest of the run: it is the one that owns its direction's bytes)
        enc_k = max((k for k in rf["by_kernel"] if k["kernel"].startswith("enc:") and k.get("owns_direction_bytes")), key=lambda k: k["ms"], default=None)
        if enc_k and rf["by_direction"]["encode"]["ms"] > 0:
            share = enc_k["ms"] / rf["by_direction"]["encode"]["ms"]
            filled = result["encode_only"]["ms_per_frame"] * share
            rf["solo_vs_filled"] = {"kernel": enc_k["kernel"], "solo_ms": enc_k["ms"], "filled_ms_per_frame": round(filled, 4),
                                    "solo_frac": enc_k.get("frac"),
                                    "filled_frac": round(enc_k["algorithmic_bytes"] / (filled * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "note": "filled = encode-only ms per frame with four pipelines x the kernel's share of the encoder's solo GPU time"}
        full = {}
        for mode in ("encode", "decode", "both"):
            m = measure(lib, spec, device, dev_index, barrier, mode=mode, streams=args.streams, steps=3, warmup=1, min_seconds=0.3, host_io=True)
            full[mode if mode != "both" else "encode_decode"] = brief(spec, m)
        full["note"] = ("host buffers in and out (pinned; GPUJPEG_ENCODER_INPUT_IMAGE / enc_out_val_pinned, host JPEG in / CUSTOM_BUFFER out): what a drop-in "
                        "caller of the reference API sees, PCIe transfers included; raw images cross the link on the library's one stream per direction "
                        "(copy lanes, DESIGN 4.6: both directions at once; the developer setting GJ_COPY_LANES=0 gives every coder's copies its own stream as before round 5)")
        result["full_api"] = full
    if not args.no_workloads and args.workload == "8k" and args.mode == "both":
        table = {"8k": {"mpix_s": result["value"], "ms_per_frame": round(result["ms_per_step"] / result["config"]["frames_per_step_per_gpu"], 4),
                        "jpeg_bytes": result["config"]["jpeg_bytes"], "streams": result["config"]["streams_per_gpu"]}}
        # sizes with the headline's content, then the headline's size with the reference's own contents (`.tst` noise and
        # gradient, src/utils/image_delegate.c:562-603, and its camera sample) -- each with the same roofline statement as the headline:
        # algorithmic bytes (raw + JPEG) per launch against 8 TB/s for the dominant kernel and for each direction (durations: hipEvents
        # with one pipeline, the GPU otherwise idle)
        for key, name, pattern in (("hd", "hd", args.pattern), ("4k", "4k", args.pattern), ("16k", "16k", args.pattern), ("16k422", "16k422", args.pattern),
                                   ("8k_noise", "8k", "noise"), ("8k_gradient", "8k", "gradient"), ("8k_camera", "8k", "camera")):
            sp = Spec(lib, name, pattern, args.quality, device, 12345)
            m = measure(lib, sp, device, dev_index, barrier, mode="both", streams=args.streams, steps=5, warmup=2, min_seconds=0.3, want_solo=True)
            alg = sp.raw_bytes + m["jpeg_bytes"]
            nblk = ((sp.width + 7) // 8) * ((sp.height + 7) // 8) * (2 if sp.is422 else 3)
            tokm = nblk >= (900000 if sp.is422 else 300000) and m["jpeg_bytes"] <= (12 if sp.is422 else 16) * nblk and not _TUNE["no_tokens"]
            nm = kernel_names(sp, m["solo_ms"], tokm)
            solo_ = m["solo_ms"]
            live_ = [i for i in range(9) if solo_[i] > 0.006]
            dom_ = max(live_, key=lambda i: solo_[i])
            tr_ = load_traffic("kernels", key) if pattern == "natural" else {}
            fr = lambda ms: round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else None  # noqa: E731
            enc_ms_, dec_ms_ = float(sum(solo_[i] for i in live_ if i < 5)), float(sum(solo_[i] for i in live_ if i >= 5))
            # the kernel with the longest duration; its HBM fraction is its own when it moves its direction's bytes alone (the fused encoders), its
            # DIRECTION's otherwise (an entropy decoder never sees the raw pixels: pricing it with them crowned it "dominant" in round 4)
            own_ = owns_direction(nm[dom_])
            basis_ms = float(solo_[dom_]) if own_ else (enc_ms_ if dom_ < 5 else dec_ms_)
            one_way = {}
            if key in ("hd", "4k", "16k", "16k422"):  # the BASELINE configurations: each direction alone as well (config 2 is HD ENCODE: four pipelines, device resident)
                for mode in ("encode", "decode"):
                    mm = measure(lib, sp, device, dev_index, barrier, mode=mode, streams=args.streams, steps=3, warmup=1, min_seconds=0.15)
                    one_way[f"{mode}_only_mpix_s"] = brief(sp, mm)["mpix_s"]
            table[key] = dict(brief(sp, m), **one_way, workload=sp.describe(), data=DATA_NOTE[pattern].format(w=sp.width, h=sp.height),
                              solo_gpu_ms={"encode": round(enc_ms_, 4), "decode": round(dec_ms_, 4)},
                              roofline={"bound": "hbm", "kernel": nm[dom_], "ms": round(float(solo_[dom_]), 4), "frac": fr(basis_ms),
                                        "frac_basis": "the kernel's own duration" if own_ else "the duration of all kernels of its direction (a stage kernel)",
                                        "achieved": round(alg / (basis_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "algorithmic_bytes_per_launch": int(alg), "traffic": tr_.get(nm[dom_]) if own_ else None,
                                        "frac_encode_direction": fr(enc_ms_), "frac_decode_direction": fr(dec_ms_),
                                        "by_kernel": {nm[i]: round(float(solo_[i]), 4) for i in live_}})
            del sp
            torch.cuda.empty_cache()
        # the sweep the reference publishes (README.md:125-128 encode, :160-161 decode: 4K / HD / 8K at q10 ... q100): the 8K frame with its camera
        # sample's contents at every quality, both directions together and each alone (four pipelines, device resident), the GPU time of each
        # direction alone and its fraction of the HBM roofline -- where between q75 and noise does the decoder leave token mode, and what does it cost?
        sweep = {}
        for q in (10, 20, 30, 40, 50, 60, 70, 80, 90, 100):
            sp = Spec(lib, "8k", "camera", q, device, 12345)
            m = measure(lib, sp, device, dev_index, barrier, mode="both", streams=args.streams, steps=3, warmup=1, min_seconds=0.15, want_solo=True)
            alg = sp.raw_bytes + m["jpeg_bytes"]
            solo_ = m["solo_ms"]
            live_ = [i for i in range(9) if solo_[i] > 0.006]
            enc_ms_, dec_ms_ = float(sum(solo_[i] for i in live_ if i < 5)), float(sum(solo_[i] for i in live_ if i >= 5))
            fr = lambda ms: round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else None  # noqa: E731
            row = {"mpix_s": brief(sp, m)["mpix_s"], "jpeg_bytes": m["jpeg_bytes"], "solo_gpu_ms": {"encode": round(enc_ms_, 4), "decode": round(dec_ms_, 4)},
                   "frac_encode_direction": fr(enc_ms_), "frac_decode_direction": fr(dec_ms_)}
            for mode in ("encode", "decode"):
                mm = measure(lib, sp, device, dev_index, barrier, mode=mode, streams=args.streams, steps=3, warmup=1, min_seconds=0.12)
                row[f"{mode}_only_mpix_s"] = brief(sp, mm)["mpix_s"]
            sweep[f"q{q}"] = row
            del sp
            torch.cuda.empty_cache()
        table["8k_camera_quality_sweep"] = dict(sweep, workload="7680x4320 RGB 4:4:4, the reference's camera sample tiled, non-interleaved, restart auto; quality 10 ... 100",
                                                 note="the reference's tables: README.md:125-128 (encode), :160-161 (decode); mpix_s = encode + decode of every frame, "
                                                      "four pipelines; *_only = one direction alone, device resident")
        ba = argparse.Namespace(**vars(args))
        ba.batch, ba.workload, ba.steps, ba.warmup, ba.verify = 256, "4k", 2, 1, False
        b = run_batch(ba, lib, device, dev_index, 0, 1, *WORKLOADS["4k"], emit=False)
        table["batch256_4k"] = {"frames_s": b["value"], "mpix_s": b["mpix_s"], "ms_per_step": b["ms_per_step"], "workload": b["config"]["workload"]}
        torch.cuda.empty_cache()
        # the same batch, and one of 256 HD frames, through the batch calls (every kernel once per chunk of frames) and, for HD, frame by frame
        for key, wl, api in (("batch256_4k_batched", "4k", "batch"), ("batch256_hd", "hd", "frame"), ("batch256_hd_batched", "hd", "batch"),
                             ("batch256_hd422_batched", "hd422", "batch")):
            ba.workload, ba.batch_api = wl, api
            ba.steps, ba.warmup = 6, 2  # (a pass over 256 HD frames is 3.5 ms: two passes are over before buffers and clocks have settled)
            b = run_batch(ba, lib, device, dev_index, 0, 1, *WORKLOADS[wl], emit=False)
            table[key] = {"frames_s": b["value"], "mpix_s": b["mpix_s"], "ms_per_step": b["ms_per_step"], "workload": b["config"]["workload"], "api": b["config"]["api"]}
            torch.cuda.empty_cache()
        ba.workload, ba.batch_api, ba.steps, ba.warmup = "4k", "frame", 2, 1
        ba.batch_io = "host"  # the same batch from pinned host memory in and out (6.4 GB each way per pass over PCIe)
        b = run_batch(ba, lib, device, dev_index, 0, 1, *WORKLOADS["4k"], emit=False)
        table["batch256_4k_host"] = {"frames_s": b["value"], "mpix_s": b["mpix_s"], "ms_per_step": b["ms_per_step"], "workload": b["config"]["workload"],
                                     "data": b["data"]}
        torch.cuda.empty_cache()
        result["workloads"] = table
    if not args.no_cpu_baseline:  # reported baselines, on the host cores of rank 0 at N = 1 only
        host_frame = spec.frame.cpu().numpy().reshape(-1)
        result["cpu_baseline"] = cpu_baseline(spec, host_frame)
        ac = cpu_baseline_all_cores(spec, host_frame)
        if ac:
            result["cpu_baseline_all_cores"] = ac


if __name__ == "__main__":
    main()
