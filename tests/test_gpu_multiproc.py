"""-m gpu: the multi-GPU paths with real codec work in more than one process / thread.

The GPU box of the test tier has ONE MI355X, so both proofs run their ranks / devices on device 0 (bench.py maps rank r to device
r mod device_count and then carries its barrier and timing reduction over gloo; tools/mgpu_encode.c maps device d to d mod
device_count). What is checked is what an 8-GPU run relies on: the shards cover the batch exactly once, every rank's frames are
bit-exact against the oracle while another rank codes at the same time, and the reductions produce one whole-job line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, args, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints exactly one line
    return json.loads(lines[0])


def test_two_ranks_shard_a_batch_bit_exact(gpu_lib):
    """BASELINE config 5 in small: 16 HD frames (seeds 12345 + i) over 2 ranks, every frame of every shard checked against the oracle."""
    d = _torchrun(2, ["--gpus", "2", "--batch", "16", "--workload", "hd", "--steps", "2", "--warmup", "1", "--verify"])
    assert d["n_gpus"] == 2 and d["config"]["frames_total"] == 16 and d["config"]["frames_per_gpu"] == 8
    assert d["verified_bit_exact"] is True
    assert d["scaling"] == "strong" and d["value"] > 0 and d["psnr_last_frame_db"] > 30


def test_two_ranks_shard_a_batch_through_the_batch_calls(gpu_lib):
    """the same shards through gpujpeg_amd_encoder_encode_batch / gpujpeg_amd_decoder_decode_batch (every kernel once per chunk of frames): every
    stream and every decoded frame of both ranks against the oracle, and the line says that the batched launches did the work"""
    d = _torchrun(2, ["--gpus", "2", "--batch", "16", "--workload", "hd", "--steps", "2", "--warmup", "1", "--verify", "--batch-api", "batch",
                      "--batch-streams", "1"])
    assert d["n_gpus"] == 2 and d["config"]["frames_total"] == 16 and d["config"]["frames_per_gpu"] == 8
    assert d["verified_bit_exact"] is True
    assert "((8, 0), (8, 0))" in d["config"]["api"], d["config"]["api"]
    assert d["value"] > 0 and d["psnr_last_frame_db"] > 30


@pytest.mark.parametrize("workload,frames", [("4k", 10), ("hd422", 20)])
def test_batch_calls_at_full_size_bit_exact(gpu_lib, workload, frames):
    """BASELINE config 5's frames (4K RGB) and HD packed 4:2:2 frames through the batch calls with everything resident in HBM, one process:
    every stream and every decoded frame against the oracle (bench.py --verify), all of them coded by the batched launches"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(frames), "--workload", workload, "--batch-api", "batch",
                        "--batch-streams", "1", "--steps", "1", "--warmup", "1", "--verify"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["verified_bit_exact"] is True and d["config"]["frames_total"] == frames
    assert f"(({frames}, 0), ({frames}, 0))" in d["config"]["api"], d["config"]["api"]


def test_two_ranks_headline_line(gpu_lib):
    """the weak-scaling headline path at N = 2: one JSON line, whole-job throughput over both ranks, per-rank frames of their own seed"""
    d = _torchrun(2, ["--gpus", "2", "--workload", "hd", "--steps", "3", "--warmup", "1", "--min-seconds", "0.1", "--lean", "--verify"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["verified_bit_exact_encode"] is True and d["verified_bit_exact_decode"] is True
    assert d["roofline"]["frac"] > 0


def test_mgpu_encode_threads_and_pinned_staging(gpu_lib):
    """tools/mgpu_encode.c: one host thread per coder, pinned staging, 2 coders per device, 2 (virtual) devices, encode + decode of 48 HD
    frames; equal frames must give equal streams on every coder."""
    exe = os.path.join(ROOT, "gpujpeg_amd", "lib", "mgpu_encode")
    assert os.path.exists(exe), "built by gpujpeg_amd/csrc/Makefile"
    r = subprocess.run([exe, "48", "1920", "1080", "2", "2", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ok"] and d["frames"] == 48 and d["streams_consistent"] and d["coders_per_device"] == 2
    assert d["digest_comparisons"] >= 8  # (the consistency check really compared streams of equal frames between coders)


def test_mgpu_encode_batch_calls(gpu_lib):
    """the same tool with every coder handing 8 frames at a time to the batch calls (C caller of include/gpujpeg_amd_ext.h, pinned host memory in
    and out): equal frames still give equal streams on every coder"""
    exe = os.path.join(ROOT, "gpujpeg_amd", "lib", "mgpu_encode")
    r = subprocess.run([exe, "64", "1920", "1080", "2", "2", "1", "1", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ok"] and d["frames"] == 64 and d["batch"] == 8 and d["streams_consistent"] and d["digest_comparisons"] >= 8


def test_bench_self_launches_its_ranks(gpu_lib):
    """`python bench.py --gpus 2` as ONE process (the driver's command shape, no torchrun in front): bench.py becomes the launcher, two
    ranks run, and the line says so."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "hd", "--steps", "3", "--warmup", "1",
                        "--min-seconds", "0.1", "--lean"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["frames_all_ranks"] > 0 and d["value"] > 0
