"""bench.py -- whole-job throughput of the mPLUG-Video pre-train step on N MI355X GPUs.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: mPLUG-Video GPT3-1.3B pre-train step (freezeGPT recipe,
TimeSformer/CLIP-B16 vision tower -- SURVEY.md R4), per-GPU batch 32 x 8 frames x 224^2 + 32-token
titles, bf16, synthetic data, random-init weights.  A step = forward + backward + DP gradient
all-reduce (overlapped) + global-norm clip + AdamW, dropout live (train mode).  One JSON line is
printed by rank 0 (contract in the task statement) with `roofline` (bf16 MFMA GEMM family, timed
live with HIP events on the launch stream) and `cpu_baseline` (the oracle restatement of the
reference path timed on the host cores; reported, not the target).
"""
import argparse
import json
import math
import os
import sys
import time
import types

# the host driver of this pool only supports dmabuf IPC: without it RCCL across processes fails in hipIpcGetMemHandle.  The launch environment
# exports it already; set here too (before the HIP runtime comes up) so that a bare `torchrun bench.py --gpus N` is enough
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PORT_VS_REFERENCE_STEP_TIME = 0.74      # the oracle port's CPU step time / the reference modules' (profiles/r03_cpu_baseline_calibration.txt)


class Shapes:
    """1.3B recipe dims (configs/models/config_gpt3_1.3B.json, clip-b16.json with num_frames 8)."""
    img_size, patch_size, vit_dim, vit_depth, vit_heads, vit_mlp_ratio, num_frames = 224, 16, 768, 12, 8, 4, 8
    vit_ln_eps, num_queries = 1e-6, 128
    hidden, layers, heads, ffn, vocab, max_pos, gpt_ln_eps = 2048, 24, 32, 8192, 51200, 2048, 1e-5


class ShapesD(Shapes):
    """2.7B decoder dims (BASELINE.json configs[3]: hidden 2560, 32 layers, head_dim 80) -- `--config D`, a side line only."""
    hidden, layers, heads, ffn = 2560, 32, 32, 10240


# `--config Y`: the geometry the reference SHIPS (configs/pretrain/gpt3_1.3B/pretrain_gpt3_freezeGPT_youku_v0.yaml:18-25: per-GPU batch 48,
# num_frames 4, max_length 80 -> S = 128 + 80 = 208), 1.3B dims.  A side line: the headline stays BASELINE.json configs[1] (config B).
GEOMETRY = {"B": dict(batch=32, frames=8, text_len=32), "D": dict(batch=16, frames=8, text_len=32),
            "E": dict(batch=96, frames=16, text_len=32),      # configs/retrieval/retrieval_gpt3_1.3B_youku_v0.yaml:19,24
            "Y": dict(batch=48, frames=4, text_len=80)}


def gemm_source_digest():
    """Content hash of the GEMM kernel sources: a PMC traffic figure is only quoted for the kernels it was measured on."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "youku-mplug_amd", "csrc")
    for name in ("gemm.hip", "gemm256.hip", "gemm_args.h", "mpv_common.h"):
        with open(os.path.join(base, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def algorithmic_train_flops(B, T, L, s=Shapes):
    """forward FLOPs (mul-add = 2), x3 for trainable parts, x2 for the frozen GPT.  EXECUTED work:
    temporal_attn.proj + temporal_fc run as ONE composed projection in both directions (one token-row product forward, one dgrad
    and one wgrad plus three D^3 weight products per block instead of two token-row products of each kind), the LM head and the
    top decoder layer's row-wise part run on the loss window."""
    D, N, Q, H, V, Lyr = s.vit_dim, (s.img_size // s.patch_size) ** 2, s.num_queries, s.hidden, s.vocab, s.layers
    M = B * T * N
    S = Q + L
    vit = 2 * M * D * D + s.vit_depth * (M * 2 * D * (2304 + 768 + 2304 + 768) + 2 * D * D * D + (M + B) * 4 * D * 4 * D
                                         + B * T * 2 * D * 4 * D + B * T * 8 * 4 * (N + 1) ** 2 * (D // s.vit_heads) * s.vit_heads / 8
                                         + B * N * 8 * 4 * T * T * (D // s.vit_heads) * s.vit_heads / 8)
    Sk = 1 + T * N
    pool = 2 * B * Q * D * D + 2 * B * Sk * D * 2 * D + 4 * B * Q * (Sk + 1) * D + 2 * B * Q * D * D + 4 * B * Q * D * 4 * D
    fc = 2 * B * Q * D * H
    # LM head on the L text positions only: the Q query slots are always masked (distributed_gpt3.py:142-159), their logits
    # feed nothing (the reference evaluates them anyway: 2*B*S*H*V); counted as executed here, so step_frac is not inflated
    # likewise the top decoder layer: its projection and MLP run on the L text rows only (the other rows of the last hidden state
    # feed nothing once the LM head is windowed); qkv and the attention of that layer stay whole
    gpt = ((Lyr - 1) * 2 * B * S * H * (3 * H + H + 2 * s.ffn) + 2 * B * S * H * 3 * H + 2 * B * L * H * (H + 2 * s.ffn)
           + Lyr * 4 * B * S * S * H + 2 * B * L * H * V)
    return 3.0 * (vit + pool + fc) + 2.0 * gpt


def algorithmic_train_flops_E(B, T, L, s=Shapes, E=256):
    """BASELINE.json configs[4] (ITC retrieval step, models/distributed_gpt3.py:938-980): the video tower trains (x3, with the
    composed temporal-projection backward counted as executed), the frozen text tower runs FORWARD only on the L title tokens
    (nothing trainable sits below its hidden state), plus the two projection heads and the similarity products."""
    D, N, H, Lyr = s.vit_dim, (s.img_size // s.patch_size) ** 2, s.hidden, s.layers
    M = B * T * N
    vit = 2 * M * D * D + s.vit_depth * (M * 2 * D * (2304 + 768 + 2304 + 768) + 2 * D * D * D + (M + B) * 4 * D * 4 * D
                                         + B * T * 2 * D * 4 * D + B * T * 4 * (N + 1) ** 2 * D + B * N * 4 * T * T * D)
    gpt = Lyr * 2 * B * L * H * (3 * H + H + 2 * s.ffn) + Lyr * 4 * B * L * L * H
    heads = 3.0 * (2 * B * D * E + 2 * B * H * E) + 3.0 * 2 * 2 * B * B * E
    return 3.0 * vit + gpt + heads


# `--_test-cpu` is a TEST HOOK for the N > 1 control flow (tests/test_pipeline_cpu.py::test_bench_control_flow_world8_gloo): eight gloo
# ranks run this file's barriers / MAX-reduce / rank-0-only JSON line on the torch stand-ins of tests/standin_ops.py.  It is a command-line
# flag, not an environment variable (a leaked variable must never turn the scoreboard program into a stand-in run: round 5's
# MPV_BENCH_DEVICE is refused below), and the line it prints is stamped `"data": "TEST ..."`.  The product has no CPU path: without the
# stand-ins installed every op raises on a CPU tensor.  Events / sync shims for that mode:
_ON_CPU = False


class _HostEvent:
    def __init__(self, enable_timing=True):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def _event():
    return _HostEvent() if _ON_CPU else torch.cuda.Event(enable_timing=True)


def _sync():
    if not _ON_CPU:
        torch.cuda.synchronize()


class GemmTimer:
    """Times every mpv_gemm_bf16 launch with HIP events on the launch stream (torch's current stream)."""

    def __init__(self):
        self.records = []
        self.shapes = []
        self.orig = None

    def __enter__(self):
        from youku_mplug_amd import ops
        self.ops, self.orig = ops, ops.gemm

        def timed(a, b, M, N, K, **kw):
            s, e = _event(), _event()
            s.record()
            r = self.orig(a, b, M, N, K, **kw)
            e.record()
            # kernel variant: <0,0> k-contiguous operands (forward, and dgrad against the frozen decoder's transposed weight
            # copies), <0,1> dgrad against trainable weights, <1,1> wgrad
            kind = "gemm<1,1> wgrad" if kw.get("trans_a") else ("gemm<0,1> dgrad" if kw.get("trans_b") else "gemm<0,0> fwd + frozen-weight dgrad")
            # algorithmic bytes of the launch: both operands + the output, plus every [M, N] tensor its epilogue reads or writes besides
            # C (the pre-activation copy of fc1, the residual, the GELU' pre-activation of a dgrad) -- all bf16.  Split-K partials
            # of the wgrad form are NOT algorithmic (they are the implementation's own traffic and show up in `traffic` only).
            extra = sum(1 for k in ("preact_out", "residual", "act_bwd_z") if torch.is_tensor(kw.get(k)))
            self.records.append((kind, 2.0 * M * N * K, s, e, 2.0 * (M * K + N * K + (1 + extra) * M * N)))
            epi = "+".join(k for k in ("bias", "act", "preact_out", "residual", "act_bwd_z", "dropout_p", "colsum_out", "row_tap_out", "kmap", "amap")
                           if (torch.is_tensor(kw.get(k)) or kw.get(k) not in (None, 0, 0.0, False, (0, 0, 0))))
            self.shapes.append((kind.split()[0], M, N, K, epi))
            return r
        ops.gemm = timed
        import youku_mplug_amd.vision as v, youku_mplug_amd.gpt3 as g, youku_mplug_amd.pretrain as p
        return self

    def __exit__(self, *exc):
        self.ops.gemm = self.orig

    def by_shape(self, path, steps):
        """MPV_BENCH_BY_SHAPE=<file>: per (form, M, N, K, epilogue) launches per step, mean us and TFLOP/s inside the step."""
        _sync()
        agg = {}
        for (kind, fl, s, e, _), key in zip(self.records, self.shapes):
            a = agg.setdefault(key, [0, 0.0, fl])
            a[0] += 1
            a[1] += s.elapsed_time(e) * 1e3
        with open(path, "w") as f:
            f.write("| form | M | N | K | epilogue | launches/step | mean us | TFLOP/s | ms/step |\n|---|---|---|---|---|---|---|---|---|\n")
            for key, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"| {key[0]} | {key[1]} | {key[2]} | {key[3]} | {key[4] or 'plain'} | {n / steps:g} | {us / n:.1f} | {fl / (us / n) / 1e6:.0f} | {us / steps / 1e3:.2f} |\n")

    def summary(self):
        _sync()
        tot = {}
        for kind, fl, s, e, by in self.records:
            t = tot.setdefault(kind, [0.0, 0.0, 0, 0.0])
            t[0] += fl
            t[1] += s.elapsed_time(e) * 1e-3
            t[2] += 1
            t[3] += by
        return tot


def pmc_traffic():
    """(HBM bytes per GEMM launch, provenance) from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled per the
    gfx950 note in MI355X_MICROARCH.md, + WRITE_SIZE; tools/profile_round.sh + tools/rocpd_pmc.py write the file).  The file records the
    digest of the GEMM sources it was measured on: a figure from other kernels is refused (None), as is a missing profile."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_gemm_latest.json")
    try:
        rec = json.load(open(path))
        if rec.get("gemm_src_sha") != gemm_source_digest():
            return None, f"profiles/pmc_gemm_latest.json was measured on other GEMM sources ({rec.get('gemm_src_sha')}): re-run tools/profile_round.sh"
        return float(rec["hbm_mb_per_launch"]) * 1e6, rec.get("source", "profiles/pmc_gemm_latest.json")
    except (OSError, ValueError, KeyError):
        return None, None


class ClockPowerSampler:
    """Shader clock (MHz) and package power (W) of this rank's GPU, sampled at ~10 Hz by a host thread while the timed region runs
    (a 74 vs 77 ms box must be explainable from the line itself).  Sources, in order: the amdsmi python binding
    (gpu_metrics: current_gfxclk / current_socket_power), the amdgpu hwmon files (freq1_input / power1_average), else nothing is
    reported (nulls).  The peak the roofline is priced against stays the guide's 2 500 TFLOP/s whatever the clock read here."""

    def __init__(self, index=0, hz=10.0):
        import threading
        self.period, self.samples, self.source = 1.0 / hz, [], None
        self._stop, self._thread = threading.Event(), None
        self.bdf = self._device_bdf(index)
        self._read = self._probe(index)

    @staticmethod
    def _device_bdf(index):
        """PCI address (domain:bus:device) of HIP device `index`.  Neither amdsmi's handle order nor the lexicographic order of
        /sys/class/drm/card* is the HIP device order (card10 sorts before card2; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES remap),
        so the sampler finds ITS device by bus address; only when the address cannot be read does it fall back to the index."""
        try:
            pr = torch.cuda.get_device_properties(index)
            return f"{int(pr.pci_domain_id):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}"
        except Exception:
            return None

    def _probe(self, index):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
            h = None
            if self.bdf is not None:
                for cand in handles:
                    try:
                        if str(amdsmi.amdsmi_get_gpu_device_bdf(cand)).lower().startswith(self.bdf):
                            h = cand
                            break
                    except Exception:
                        pass
            if h is None:
                h = handles[index]

            def num(v):
                return float(v) if isinstance(v, (int, float)) and 0 < float(v) < 65535 else None

            def read():
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                clk = num(m.get("current_gfxclk"))
                if clk is None:
                    xs = [num(x) for x in (m.get("current_gfxclks") or [])]
                    xs = [x for x in xs if x]
                    clk = sum(xs) / len(xs) if xs else None
                return clk, num(m.get("current_socket_power")) or num(m.get("average_socket_power"))
            if read() != (None, None):
                self.source = "amdsmi gpu_metrics"
                return read
        except Exception:
            pass
        try:
            import glob
            cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            hw = None
            if self.bdf is not None:
                for c in cands:        # .../cardK/device -> /sys/devices/pci.../<domain:bus:device.function>
                    if os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(c)))).lower().startswith(self.bdf):
                        hw = c
                        break
            if hw is None:
                hw = cands[index]

            def read():
                def f(name, scale):
                    try:
                        return float(open(os.path.join(hw, name)).read()) / scale
                    except (OSError, ValueError):
                        return None
                return f("freq1_input", 1e6), f("power1_average", 1e6) or f("power1_input", 1e6)
            if read() != (None, None):
                self.source = "amdgpu hwmon"
                return read
        except Exception:
            pass
        return None

    def __enter__(self):
        if self._read is not None:
            import threading

            def loop():
                while not self._stop.is_set():
                    try:
                        self.samples.append(self._read())
                    except Exception:
                        pass
                    self._stop.wait(self.period)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)

    def summary(self):
        def med(xs):
            xs = sorted(x for x in xs if x is not None)
            return round(xs[len(xs) // 2], 1) if xs else None
        clk, pw = [s[0] for s in self.samples], [s[1] for s in self.samples]
        return {"sclk_mhz": med(clk), "power_w": med(pw), "sclk_mhz_min": round(min((c for c in clk if c), default=0), 1) or None,
                "power_w_max": round(max((p for p in pw if p), default=0), 1) or None, "clock_power_samples": len(self.samples),
                "clock_power_source": self.source, "clock_power_device_bdf": self.bdf}


class _StdoutToStderr:
    """RCCL prints a version banner to the C stdout when its communicator comes up; rank 0's stdout must carry exactly
    one JSON line, so the banner is steered to stderr (fd-level, restored afterwards)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:        # the banner sits in the C library's stdout buffer (fully buffered on a pipe): without this it would surface at
            import ctypes      # process exit, on the restored fd 1, AFTER the JSON line
            ctypes.CDLL(None).fflush(None)
        except (OSError, AttributeError):
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)


def host_cores():
    """Cores this process may actually use: min(affinity, cgroup cpu quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(threads):
    """Oracle restatement (a port of the reference path, oracle/restate.py) on the host cores: full steps of the BENCHMARKED
    workload's per-sample shape (config B: 1.3B dims, 8 frames x 224^2, 32-token titles; bf16; forward + backward + AdamW) at a
    bounded batch of 2 clips, for about 12 s."""
    from oracle import restate
    from oracle.weights import CONFIG_B, state_dict_spec
    torch.set_num_threads(threads)
    cfg = CONFIG_B
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shape, kind in state_dict_spec(cfg):
        t = torch.empty(shape, dtype=torch.bfloat16)
        if kind == "ln_w":
            t.fill_(1.0)
        else:
            t.normal_(0.0, 0.02, generator=g)
        sd[k] = t
    trainable = [k for k in sd if not k.startswith("text_decoder.")]
    for k in trainable:
        sd[k].requires_grad_(True)
    video = torch.randn(2, 3, cfg.num_frames, 224, 224, generator=g).bfloat16()
    ids = torch.randint(0, cfg.vocab, (2, 32), generator=g)
    mask = torch.ones(2, 32, dtype=torch.long)
    state = {k: (sd[k].detach().float(), torch.zeros_like(sd[k], dtype=torch.float32), torch.zeros_like(sd[k], dtype=torch.float32))
             for k in trainable}
    def one_step(step):
        for k in trainable:
            sd[k].grad = None
        out = restate.pretrain_forward(video, ids, mask, sd, cfg)
        out["loss"].backward()
        with torch.no_grad():
            for k in trainable:
                p, m, v = state[k]
                restate.adamw_step(p, sd[k].grad.float(), m, v, step, 1e-4, 0.9, 0.999, 1e-6, 0.05)
                sd[k].copy_(p)

    one_step(1)                                  # untimed: first-touch / thread-pool warm-up
    t0, n = time.time(), 0
    while n < 24 and (n < 2 or time.time() - t0 < 12.0):     # a bounded sample: >= 12 s of CPU work, at most 24 steps
        n += 1
        one_step(1 + n)
    dt = (time.time() - t0) / n
    # the port's step takes 0.74x the time of the reference's own modules on the same cores (profiles/r03_cpu_baseline_calibration.txt,
    # tools/cpu_baseline_calibration.py: build container, config A): what the REFERENCE would read here is value x 0.74
    return {"value": round(2.0 / dt, 5), "value_reference_equiv": round(2.0 / dt * PORT_VS_REFERENCE_STEP_TIME, 5),
            "value_reference_equiv_source": f"value x {PORT_VS_REFERENCE_STEP_TIME} (profiles/r03_cpu_baseline_calibration.txt: port step time / reference-module step time)",
            "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n} full steps of the config-B per-sample shape at batch 2 (T=8, L=32, 1.3B dims, bf16; fwd + bwd + AdamW) = {dt * n:.1f} s on {threads} threads, "
                      f"{dt:.2f} s per step; calibration against the reference's own modules (build container, config A, same cores): the port's "
                      "step takes 0.74x the reference modules' (profiles/r03_cpu_baseline_calibration.txt), i.e. this figure flatters the CPU by ~1.35x"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["B", "D", "E", "Y"], default="B",
                    help="B = the headline 1.3B config; D = 2.7B decoder dims (side line); E = ITC retrieval fine-tune step at 16 frames (side line); "
                         "Y = the shipped pre-train YAML's geometry, per-GPU 48 x 4 frames x 80 tokens (side line)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--text-len", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--step-mode", choices=["auto", "eager", "graph"], default=None,
                    help="auto (default): on N > 1 ranks a start-up self-check compares the replayed step (a chain of HIP-graph segments, "
                         "all-reduces between them) with the eager step bit for bit on every rank; if it passes, four fenced steps of each mode "
                         "are timed and the faster one (slowest rank) runs; N = 1 runs eager.  MPV_GRAPH=1 / 0 in the environment mean "
                         "graph / eager when the flag is absent")
    ap.add_argument("--_test-cpu", dest="test_cpu", action="store_true", help=argparse.SUPPRESS)
    # TEST HOOK (tests/test_entrypoint_gpu.py): N ranks on ONE GPU over gloo on device tensors (RCCL refuses two ranks per device) -- the whole N > 1
    # flow (self-check, mode probe, fenced timed region, roofline steps on every rank) on real kernels where only 1-GPU boxes exist; the line says TEST
    ap.add_argument("--_test-one-gpu", dest="test_one_gpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    global _ON_CPU
    _ON_CPU = bool(args.test_cpu)
    if "MPV_BENCH_DEVICE" in os.environ:
        sys.exit("bench.py: MPV_BENCH_DEVICE is no longer read (round 6): the stand-in run of the CPU suite is the --_test-cpu flag; unset the variable")

    t_start = time.perf_counter()
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if _ON_CPU:
        dev = torch.device("cpu")
    else:
        if args.test_one_gpu:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    # MPV_BENCH_FORCE_DIST=1 runs the distributed code path (RCCL init, broadcast, barriers, bucketed all-reduce, max-over-ranks
    # timing) even at world size 1 -- the only way to exercise it on a 1-GPU box
    dist_on = world > 1 or os.environ.get("MPV_BENCH_FORCE_DIST", "0") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        with _StdoutToStderr():
            from youku_mplug_amd.engine import init_process_group_for_dp      # RCCL on a high-priority stream (engine.py)
            if _ON_CPU or args.test_one_gpu:
                init_process_group_for_dp("gloo", rank=rank, world_size=world)
            else:
                init_process_group_for_dp("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()                                               # brings the communicator (and its banner) up now
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"

    import youku_mplug_amd  # noqa: F401  raises if libmpv_hip.so is missing
    from youku_mplug_amd import _lib, engine as eng
    from youku_mplug_amd.pretrain import synthetic_model
    if not _ON_CPU:
        _lib.check(_lib.lib().mpv_check_device(), "mpv_check_device")

    global Shapes
    if args.config == "D":
        Shapes = ShapesD
    geo = GEOMETRY[args.config]
    if args.batch is None:
        args.batch = geo["batch"]
    if args.frames is None:
        args.frames = geo["frames"]
    if args.text_len is None:
        args.text_len = geo["text_len"]
    Shapes.num_frames = args.frames
    torch.manual_seed(1234 + rank)                                   # run_pretrain_distributed_gpt3.py:210 (initialize() broadcasts rank 0's weights)
    if args.config == "E":
        from youku_mplug_amd.retrieval import synthetic_retrieval_model
        model = synthetic_retrieval_model(Shapes, device=dev, num_frames=args.frames)
    else:
        model = synthetic_model(Shapes, device=dev, num_frames=args.frames)
    with torch.no_grad():                                            # module-default init zeroes the temporal branch
        for blk in model.visual_encoder.blocks:
            blk.temporal_fc.weight.normal_(0, 0.015)
        model.visual_encoder.temporal_embed.normal_(0, 0.015)
    model.train()
    groups = eng.get_parameter_groups(model, 0.05, model.no_weight_decay(), visual_backbone_scale=True)
    engine, opt, _, _ = eng.initialize(model=model, model_parameters=groups,
                                       config=dict(lr=1e-4, opt_betas=(0.9, 0.999), opt_eps=1e-6, clip_grad=3.0))
    if dist_on and world == 1:
        engine.reducer.always = True
    B, T, L = args.batch, args.frames, args.text_len
    video = torch.randn(B, 3, T, Shapes.img_size, Shapes.img_size, device=dev).to(torch.bfloat16)
    ids = torch.randint(0, Shapes.vocab, (B, L), device=dev)
    text = types.SimpleNamespace(input_ids=ids, attention_mask=torch.ones(B, L, dtype=torch.long, device=dev))
    total = args.warmup + args.steps
    lr_sched = [1e-4 * min(1.0, (i + 1) / 2000.0) for i in range(2 * total + 8)]     # linear warm-up (utils.py:350-372)

    idx = torch.arange(B, device=dev) + rank * B                      # retrieval: one positive per (video, title) pair across the global batch
    flops_fn = algorithmic_train_flops_E if args.config == "E" else algorithmic_train_flops

    # Step mode.  The replayed step (engine.graph_step: one HIP graph; data parallel: a chain of graph segments with the bucket
    # all-reduces issued eagerly between them) does the same work bit for bit and costs the host 3.5-5.5 ms of CPU per step instead
    # of ~31 -- which is what eight ranks sharing one host need.  It is only USED after engine.graph_self_check has shown, on this
    # job's own ranks and communicator, three replayed steps bit-identical (losses, parameters) to three eager steps from the same
    # state; the state is rewound afterwards, so both modes time the same steps.
    mode = args.step_mode or {"1": "graph", "0": "eager"}.get(os.environ.get("MPV_GRAPH", ""), "auto")
    if _ON_CPU:
        mode = "eager"

    def set_lr(i):
        for g in opt.param_groups:                                   # run_pretrain_distributed_gpt3.py:88-96
            g["lr"] = lr_sched[i] * g["lr_scale"]
    inputs = (video, text, idx) if args.config == "E" else (video, text)
    self_check = None
    use_graph = False
    if mode == "graph" or (mode == "auto" and dist_on):
        roomy = True
        if mode == "auto" and not _ON_CPU:
            # a captured step keeps a second copy of the step's activations in the graph's private pool: only where that fits easily
            # (config E at its YAML batch holds 130 GiB of activations: the replay is not even tried there)
            snap0 = engine.snapshot_state()
            set_lr(0)
            l0 = engine(*inputs)
            engine.backward(l0[0] if isinstance(l0, (tuple, list)) else l0)
            engine.step()
            _sync()
            engine.restore_state(snap0)
            peak = torch.cuda.max_memory_allocated(dev)
            roomy = peak < 0.35 * torch.cuda.get_device_properties(dev).total_memory
            if not roomy:
                self_check = {"passed": False, "detail": f"not tried: an eager step peaks at {peak / 2**30:.0f} GiB, a captured one would hold as much again"}
        if roomy:
            ok, why = engine.graph_self_check(*inputs, before_step=set_lr)
            self_check = {"passed": bool(ok), "detail": why}
            use_graph = bool(ok)

    def step(i, eager=False, replay=False):
        set_lr(i)
        if replay or (use_graph and not eager):
            return engine.graph_step(video, text, idx) if args.config == "E" else engine.graph_step(video, text)
        if args.config == "E":
            loss = engine(video, text, idx)                          # downstream/run_retrieval_distributed_gpt3.py:137
        else:
            loss, _ = engine(video, text)
        engine.backward(loss)
        engine.step()
        return loss

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    def fence():
        if dist_on:
            dist.barrier()
        _sync()
    # auto mode on a process group, self-check passed: MEASURE which mode this host / node runs faster -- up to four fenced steps of each, the
    # slowest rank's time -- and take it.  (On a 1-GPU box the replay is ~1 % slower on the device -- the second stream of the
    # weight-gradient lane overlaps less as graph branches -- and saves ~25 ms of host CPU per step: which of the two matters depends on
    # the cores eight ranks have to share, which nobody can know before the node exists.)  The state is rewound afterwards.
    mode_probe = None
    if mode == "auto" and use_graph:
        snap = engine.snapshot_state()
        probe = {}
        for arm in ("graph", "eager"):
            step(0, eager=arm == "eager")              # (untimed: the first launch of a mode)
            fence()
            nprobe = max(2, min(4, args.steps))
            tp = time.perf_counter()
            for i in range(nprobe):
                step(1 + i, eager=arm == "eager")
            fence()
            tt = torch.tensor([time.perf_counter() - tp], device=dev, dtype=torch.float64)
            if dist_on:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            probe[arm] = tt.item() / nprobe * 1e3
        engine._graph_inflight.clear()
        engine.restore_state(snap)
        mode_probe = {k: round(v, 2) for k, v in probe.items()}
        use_graph = probe["graph"] <= probe["eager"]       # (the same numbers on every rank: MAX-reduced)
    step_mode = "graph" if use_graph else "eager"
    log("model + engine built" + ("" if self_check is None else f"; graph self-check: {self_check}; mode probe (ms/step, slowest rank): {mode_probe} -> step mode {step_mode}"))
    for i in range(args.warmup):
        loss = step(i)
    _sync()
    log("warmup done")
    # per-step HIP events on the launch stream (torch's current stream is the stream every kernel of the step is launched
    # on) give the distribution; the reported value is the whole timed region between two fences (the contract)
    marks = [_event() for _ in range(args.steps + 1)]
    sampler = ClockPowerSampler(local) if (rank == 0 and not _ON_CPU) else None
    fence()
    if sampler is not None:
        sampler.__enter__()                     # a host thread reading gpu_metrics at 10 Hz: nothing is enqueued on the device
    t0 = time.perf_counter()
    c0 = time.thread_time()
    marks[0].record()
    for i in range(args.steps):
        loss = step(args.warmup + i)
        marks[i + 1].record()
    t_enq = time.perf_counter() - t0            # host time to ENQUEUE the timed steps (no device sync inside a step)
    t_cpu = time.thread_time() - c0             # CPU time this thread spent doing it (the wall time above includes waiting on a full launch queue)
    fence()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if sampler is not None:
        sampler.__exit__()
    final_loss = loss.item()                    # of the LAST TIMED step (under graph replay `loss` is a static buffer the extra launches below overwrite)
    log(f"host enqueue time {t_enq / args.steps * 1e3:.1f} ms/step wall, {t_cpu / args.steps * 1e3:.1f} ms/step CPU (launch-bound if the CPU figure approaches the step time)" +
        (" [graph replay]" if use_graph else ""))
    # the same on an IDLE queue (nothing to wait for): what the host really spends to launch one step
    t_idle = []
    for i in range(3):
        _sync()
        t1, c1 = time.perf_counter(), time.thread_time()
        step(total - 1)
        t_idle.append((time.perf_counter() - t1, time.thread_time() - c1))
    _sync()
    log(f"host time to launch one step on an idle queue: {min(t[0] for t in t_idle) * 1e3:.1f} ms wall, {min(t[1] for t in t_idle) * 1e3:.1f} ms CPU")
    host = {"mode": step_mode, "cpu_ms_per_step": round(t_cpu / args.steps * 1e3, 2), "enqueue_wall_ms_per_step": round(t_enq / args.steps * 1e3, 2),
            "idle_queue_launch_ms": round(min(t[0] for t in t_idle) * 1e3, 2), "graph_self_check": self_check,
            "mode_probe_ms_per_step": mode_probe}
    # the OTHER mode's host cost, for the record (every rank: the steps carry the collectives): the CPU time this thread spends enqueueing
    # eight steps back to back (a filling queue, as in the timed region)
    other = None
    if self_check is not None and self_check["passed"]:
        other_eager = use_graph
        nother = max(2, min(8, args.steps))
        c1 = time.thread_time()
        for i in range(nother):
            step(total - 1, eager=other_eager, replay=not other_eager)
        other = round((time.thread_time() - c1) / nother * 1e3, 2)
        _sync()
        engine._graph_inflight.clear()
    host["cpu_ms_per_step_eager"] = host["cpu_ms_per_step"] if not use_graph else other
    host["cpu_ms_per_step_graph"] = host["cpu_ms_per_step"] if use_graph else other      # None: the replay never ran (N = 1 default, failed self-check, --step-mode eager)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if dist_on:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = dt.item()
    log(f"timed region done: {dt / args.steps * 1e3:.1f} ms/step")
    ms_ = {} if _ON_CPU else torch.cuda.memory_stats(dev)
    log(f"device memory: peak allocated {ms_.get('allocated_bytes.all.peak', 0) / 2**30:.1f} GiB, peak reserved {ms_.get('reserved_bytes.all.peak', 0) / 2**30:.1f} GiB, "
        f"allocator retries {ms_.get('num_alloc_retries', 0)}, device mallocs {ms_.get('num_device_alloc', 0)} / frees {ms_.get('num_device_free', 0)}")
    assert math.isfinite(final_loss), "non-finite loss in the timed region"

    roof = None
    if not args.no_roofline:
        # EVERY rank runs these extra steps (each holds the step's gradient all-reduces: a rank that stepped alone would leave its
        # peers' next collective unmatched); only rank 0's launch timings go into the line
        nroof = min(args.steps, 3)
        # per-launch durations are taken with the weight-gradient lane off (vision._WgradLane: in the timed region the
        # ViT wgrad GEMMs run on a second stream and share the chip with the dgrad launches, which stretches both
        # kernels' start-to-end times); tools/profile_round.sh traces the same serial mode (MPV_WGRAD_STREAM=0)
        lane = getattr(model.visual_encoder, "_wgrad_lane", None)
        lane_on = lane.on if lane is not None else False
        if lane is not None:
            lane.on = False
        with GemmTimer() as gt:
            for i in range(nroof):
                step(total + i, eager=True)
        tot = gt.summary()
        if rank == 0 and os.environ.get("MPV_BENCH_BY_SHAPE"):
            gt.by_shape(os.environ["MPV_BENCH_BY_SHAPE"], nroof)
        if lane is not None:
            lane.on = lane_on
    if rank == 0 and not args.no_roofline:
        fl = sum(v[0] for v in tot.values())
        tt = sum(v[1] for v in tot.values())
        n = sum(v[2] for v in tot.values())
        by = sum(v[3] for v in tot.values())
        traffic, traffic_src = pmc_traffic()
        step_fl = flops_fn(B, T, L, Shapes)
        step_s = dt / args.steps
        # `frac` is SURVEY section 8(d)'s quantity: algorithmic (executed) FLOPs of the step / measured step time / MFMA peak -- the
        # step-level fraction north_star's 0.40 target is stated on.  The GEMM family alone (the dominant kernels, HIP events per
        # launch) is `gemm_frac` / `gemm_achieved`.
        roof = {"bound": "mfma", "achieved": round(step_fl / step_s / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(step_fl / step_s / 1e12 / PEAK_BF16_TFLOPS, 4),
                "frac_definition": "step-level (SURVEY 8(d)): executed algorithmic FLOPs of the whole step / driver-contract step time / 2.5 PF; gemm_frac = the GEMM family's own launches",
                "step_algorithmic_tflop": round(step_fl / 1e12, 2),
                "kernel": "gemm256_kernel<TA,TB,KMAP> (256x256 eight-phase; fwd/dgrad/wgrad) + gemm_bf16_kernel (128x128 fallback)",
                "gemm_achieved": round(fl / tt / 1e12, 1), "gemm_frac": round(fl / tt / 1e12 / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "bytes per mpv_gemm_bf16 call (L2-miss side: 2*FETCH_SIZE + WRITE_SIZE; a call is 1-3 row-band launches)", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(by / n, 0),
                "measured": "HIP events around every mpv_gemm_bf16 launch in %d extra steps after the timed region, weight-gradient lane off (kernels do not overlap)" % nroof,
                "launches_per_step": n // nroof, "avg_launch_us": round(tt / n * 1e6, 1), "avg_launch_gflop": round(fl / n / 1e9, 2),
                "gemm_ms_per_step": round(tt / nroof * 1e3, 2),
                "by_kernel": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(v[1] / nroof * 1e3, 2), "launches": v[2] // nroof}
                            for k, v in tot.items()}}
        if sampler is not None:
            roof.update(sampler.summary())
    if dist_on:
        dist.barrier()
    if rank == 0:
        from youku_mplug_amd import ops as _ops
        print(f"[bench] mpv_gemm_bf16 calls in this process: {_ops.gemm_calls}", file=sys.stderr)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            ncores = host_cores()
            log(f"cpu baseline on {ncores} cores ...")
            cpu = cpu_baseline(ncores)
            log("cpu baseline done")
        names = {"B": "mPLUG-Video GPT3-1.3B pretrain step (freezeGPT, TimeSformer CLIP-B/16)",
                 "Y": "mPLUG-Video GPT3-1.3B pretrain step (freezeGPT, TimeSformer CLIP-B/16) at the shipped YAML's geometry -- side line, not the headline config",
                 "D": "mPLUG-Video GPT3-2.7B pretrain step (freezeGPT, TimeSformer CLIP-B/16) -- side line, not the headline config",
                 "E": "mPLUG-Video GPT3-1.3B ITC retrieval fine-tune step (run_retrieval_distributed_gpt3, TimeSformer CLIP-B/16) -- side line, not the headline config"}
        rec = {"metric": "video-text samples/sec/node, mPLUG-Video 1.3B pretrain step" if args.config == "B" else
               "video-text samples/sec/node (side line: " + {"D": "2.7B pretrain step)", "E": "1.3B ITC retrieval step)",
                                                              "Y": "1.3B pretrain step, shipped YAML geometry 48 x 4 frames x 80 tokens)"}[args.config],
               "value": round(world * B * args.steps / dt, 2),
               "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2),
               "ms_per_step_hip_events": {"median": round(per_step[len(per_step) // 2], 2), "mean": round(sum(per_step) / len(per_step), 2),
                                          "p10": round(per_step[len(per_step) // 10], 2), "p90": round(per_step[(9 * len(per_step)) // 10], 2)},
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "TEST (cpu stand-ins via --_test-cpu: control flow only, not a measurement)" if _ON_CPU else
               "TEST (--_test-one-gpu: every rank on one GPU over gloo, not a measurement)" if args.test_one_gpu else "synthetic",
               "step_mode": step_mode, "host": host,
               "config": {"workload": f"{names[args.config]}, per-GPU bs={B} x {T} frames x 224^2 + {L}-token titles",
                          "global_batch": world * B, "frames": T, "text_len": L, "queries": Shapes.num_queries, "parallelism": f"dp{world}",
                          "trainable_params_m": round(engine.flat.numel / 1e6, 1), "final_loss": round(final_loss, 6)},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(rec), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
