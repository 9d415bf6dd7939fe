#!/usr/bin/env python3
"""Throughput of the batched VTM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (parameter frames in HBM -> audio samples in HBM) over
one batch of synthetic utterances.

Workload.  BASELINE.json's `metric` names no configuration, so the N=1 line is the largest
single-GPU entry of `configs`: configs[3], "Batch=4096 30s long-form tracks, 2x oversampled
tube, fp32 vs fp64 sweep" = 4096 utterances x 7500 frames, VocalTractModel2<TFloat,2>
semantics (SectionDelay 2, 40 068 Hz internal), 44.1 kHz out; 1.97 GB of frames in and
21.6 GB of samples out per launch, all resident in HBM.  The headline runs it in float (the
reference's TFloat = float class, output bit-identical to it); `extras` carries the same
workload in mixed and fp64 (the sweep), 4096 x 2 s and configs[1] (256 x 2 s, VTM0) in all
three, and for each precision how far its samples are from the fp64 path's.  For N>1 the driver
starts one process per GPU (torch.distributed.run): the global batch (default 4096 per GPU =
configs[4]; `--global-batch G` fixes the total instead) is cut into contiguous shards by
gama_tts_amd.shard.shard_range, every rank synthesizes its own shard (independent utterances,
no data-path collective — SURVEY.md 8e); the only communication is the barrier and the MAX of
the elapsed time.

`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts its N ranks itself: before anything touches a GPU
the parent spawns one fresh child process per device (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
127.0.0.1 rendezvous), relays rank 0's line and exits with the worst child status -- the torch.distributed.run path is
unchanged.

Rank 0 prints one JSON line.  `roofline.achieved` = algorithmic bytes per launch (64 B per
input frame + 4 B per output sample) / the synthesis kernel's mean duration
measured with HIP events on the launch stream; `roofline.valu` prices the same launches against
the vector-ALU peak (the resource that actually binds).  `cpu_baseline` times the
real reference (oracle/_ref, compiled from /root/reference in the build container; "port" = our
C oracle when that binary is absent) on one host core over a bounded sample of the same workload, once per reference class
(float and double), with the host CPU's model and core count beside it.  `parity_check` compares two utterances of the
TIMED output buffer with the oracle after the timed region (the oracle is the checker here, never the thing measured);
`end_to_end` times the host-buffer entries (H2D frames + kernel + D2H samples through page-locked buffers, float32 and
int16 output) beside the kernel-only `value`, never instead of it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = {"f32": 157.3, "mixed": 78.6, "f64": 78.6}  # vector peaks (same guide: fp32 157.3; fp64 vector = half)

# Algorithmic flops: per internal step 367 (interpolation 16, radii 16, conversions 20, junction
# coefficients 48, glottal lookup 16 + 49-tap decimator 98, source mix 17, tube 118, throat 4, filters 14) and per OUTPUT
# sample 106 (26 taps x (coefficient interpolation 2 + multiply-add 2) + 2); transcendentals and divisions not counted.
FLOP_PER_STEP = 367.0
FLOP_PER_OUTPUT = 106.0

DEFAULT_BATCH_PER_GPU = 4096   # configs[3] / configs[4]
DEFAULT_FRAMES = 7500          # 30 s at 250 Hz
DEFAULT_DELAY = 2              # "2x oversampled tube" = VocalTractModel2<TFloat,2>


def rank_workload(args, rank, world):
    """Which utterances of the global batch this rank synthesizes: (global_batch, lo, hi).

    Contiguous shards from gama_tts_amd.shard.shard_range (the same function the gloo test drives); the global batch
    is `--global-batch` (strong: fixed total) or `--batch` per GPU x world (weak, BASELINE configs[4] = 4096 x 8)."""
    from gama_tts_amd.shard import shard_range

    per_gpu = args.batch if args.batch is not None else DEFAULT_BATCH_PER_GPU
    total = args.global_batch if args.global_batch is not None else per_gpu * world
    lo, hi = shard_range(total, rank, world)
    return total, lo, hi


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 4096, BASELINE configs[3] / [4])")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total utterances over all ranks, cut into contiguous shards (strong scaling); default --batch x ranks")
    ap.add_argument("--frames", type=int, default=None, help="control frames per utterance, 4 ms each (default 7500 = 30 s)")
    ap.add_argument("--delay", type=int, default=None, help="SectionDelay (default 2 = the 2x oversampled tube; 1 = VocalTractModel0)")
    ap.add_argument("--precision", choices=["f64", "mixed", "f32"], default="f32",
                    help="f32 (default) = the reference's TFloat = float classes (VocalTractModel0<float> = model 1, "
                         "VocalTractModel2<float,D>), output bit-identical to them; f64 = TFloat = double (models 0 / 2 / 3), "
                         "within 1e-9 of peak; mixed = fp64 with an fp32 resampler, 3e-7 of peak from the double model")
    ap.add_argument("--output-rate", type=float, default=None, help="default 44100 (48000 with --model 5, the 5_male voice's own rate)")
    ap.add_argument("--model", type=int, choices=[0, 4, 5], default=0,
                    help="0: the VocalTractModel0/2 path (default, BASELINE configs); 4: the 30+18-section tube of VocalTractModel4 "
                         "(--precision f64 = reference model 4, f32 = VocalTractModel4<float,1>; --delay ignored); 5: reference model 5 "
                         "(VocalTractModel5<double,1>, its own kernel; fp64, --delay / --precision ignored); 4 and 5 default to 256 x 500 frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU measurements (other precisions and sizes)")
    ap.add_argument("--dist-backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-buffer (H2D + kernel + D2H) measurements")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle comparison of the timed output")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU work at all: the ranks only rendezvous (gloo), cut the batch into their shards and rank 0 prints "
                         "them -- a CPU rehearsal of the launcher and the rank logic, not a measurement")
    args = ap.parse_args(argv)
    if args.model in (4, 5):
        # the other tubes keep round 1's 256 x 2 s workload (one utterance per workgroup at that size)
        if args.batch is None:
            args.batch = 256
        if args.frames is None:
            args.frames = 500
        args.delay = 1
        if args.model == 5:
            args.precision = "f64"
    if args.frames is None:
        args.frames = DEFAULT_FRAMES
    if args.delay is None:
        args.delay = DEFAULT_DELAY
    if args.output_rate is None:
        args.output_rate = 48000.0 if args.model == 5 else 44100.0
    return args


def reference_class(precision, delay, model):
    """The reference class whose output a precision reproduces, and how closely."""
    t = "float" if precision == "f32" else "double"
    if model == 5:
        cls = "VocalTractModel5<double,1> (factory model 5)"
    elif model == 4:
        cls = "VocalTractModel4<%s,1>%s" % (t, " (factory model 4)" if t == "double" else "")
    elif delay == 1:
        cls = "VocalTractModel0<%s> (factory model %d)" % (t, 1 if t == "float" else 0)
    else:
        cls = "VocalTractModel2<%s,%d>%s" % (t, delay, " (factory model 3)" if (t == "double" and delay == 3) else "")
    how = {"f32": "bit-identical float32 samples (tests/test_gpu_parity_f32.py)",
           "f64": "<= 1e-9 of peak on 2 s, 5e-9 after 30 s (tests/test_gpu_parity.py)",
           "mixed": "3e-7 of peak (fp32 resampler; tests/test_gpu_parity.py)"}[precision]
    return cls, how


def host_identity():
    """(CPU model name, logical cores) of this host from /proc/cpuinfo."""
    model, cores = None, 0
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model is None:
                    model = line.split(":", 1)[1].strip()
                if line.startswith("processor"):
                    cores += 1
    except OSError:
        pass
    return model, (cores or os.cpu_count() or 0)


def cpu_baseline(params, output_rate, delay, budget_s=6.0, float_model=False, model5=False, model4=False):
    """Single-thread CPU throughput (output samples/s) on a bounded sample of the workload."""
    import oracle
    voice = oracle.VOICE5_MALE if model5 else oracle.VOICE_MALE

    sample = params[:2]
    kind = None
    exe_kind = None
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
    except OSError:
        pass
    if oracle.ref_binary("v3") and all(x in flags for x in (" avx2", " fma", " bmi2")):
        exe_kind = "v3"
    elif oracle.ref_binary("o3"):
        exe_kind = "o3"
    model = {1: "0", 2: "2:2", 3: "3", 4: "2:4"}[delay]
    if float_model:  # the reference's TFloat = float classes
        model = {1: "1", 2: "2f:2", 3: "2f:3", 4: "2f:4"}[delay]
    if model5:
        model = "5"
    if model4:
        model = "4f" if float_model else "4"
    if exe_kind:
        kind = "reference"
        _, info = oracle.ref_synthesize(sample[0], model, output_rate, 250.0, config=voice, kind=exe_kind, repeat=1)
        per_rep = float(info["sec"])
        repeat = max(1, int(budget_s / max(per_rep, 1e-6) / len(sample)))
        total_samples = 0
        total_sec = 0.0
        for tr in sample:
            _, info = oracle.ref_synthesize(tr, model, output_rate, 250.0, config=voice, kind=exe_kind, repeat=repeat)
            total_samples += int(info["N"]) * repeat
            total_sec += float(info["sec"])
        desc = "%d utterances x %d repeats of %d frames through oracle/_ref/ref_vtm_%s (model %s)" % (
            len(sample), repeat, sample.shape[1], exe_kind, model)
    else:
        kind = "port"
        if model5:
            cfg5 = oracle.male5_config(output_rate)
            synth = lambda tr: oracle.synthesize5(cfg5, tr)[0]  # noqa: E731
        else:
            cfg = oracle.male_config(output_rate, delay, 1 if model4 else 0, float_model=int(float_model))
            synth = lambda tr: oracle.synthesize(cfg, tr)  # noqa: E731
        t0 = time.perf_counter()
        synth(sample[0])
        per = time.perf_counter() - t0
        repeat = max(1, int(budget_s / max(per, 1e-6) / len(sample)))
        total_samples = 0
        t0 = time.perf_counter()
        for tr in sample:
            for _ in range(repeat):
                total_samples += synth(tr).size
        total_sec = time.perf_counter() - t0
        desc = "%d utterances x %d repeats of %d frames through oracle/vtm_oracle.c" % (len(sample), repeat, sample.shape[1])
    cpu_model, host_cores = host_identity()
    return {"value": total_samples / total_sec, "unit": "samples/s", "cores": 1, "kind": kind, "sample": desc,
            "reference_class": reference_class("f32" if float_model else "f64", delay, 5 if model5 else (4 if model4 else 0))[0],
            "cpu_model": cpu_model, "host_cores": host_cores}


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_plan(n_ranks, argv, port):
    """The N child processes `bench.py --gpus N` starts when no launcher has: [(command, environment additions)], one per
    device, rendezvous on 127.0.0.1 (what torch.distributed.run would have exported)."""
    plan = []
    for r in range(n_ranks):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
        plan.append(([sys.executable, os.path.abspath(__file__)] + list(argv), env))
    return plan


def self_launch(args, argv, timeout_s=1500.0):
    """Start the ranks as fresh child processes (this process has not touched a GPU and never will), relay rank 0's
    output, return the worst exit status.  A rank that fails takes the others down with it (they would wait in the
    barrier forever)."""
    import subprocess

    procs = []
    for r, (cmd, env_add) in enumerate(launch_plan(args.gpus, argv, free_port())):
        env = dict(os.environ)
        env.update(env_add)
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + timeout_s
    status = [None] * len(procs)
    while any(st is None for st in status):
        for i, pr in enumerate(procs):
            if status[i] is None:
                status[i] = pr.poll()
        failed = any(st not in (None, 0) for st in status)
        if failed or time.time() > deadline:
            for i, pr in enumerate(procs):
                if status[i] is None:
                    pr.terminate()  # (our own children, by handle)
            for i, pr in enumerate(procs):
                if status[i] is None:
                    try:
                        status[i] = pr.wait(timeout=20)
                    except Exception:
                        pr.kill()
                        status[i] = pr.wait()
            if not failed:
                print("bench.py: ranks still running after %.0f s, stopped" % timeout_s, file=sys.stderr)
                return 124
            break
        time.sleep(0.05)
    return max(abs(int(st or 0)) for st in status)


def rehearse_launch(args):
    """--rehearse-launch: everything `--gpus N` does around the GPU work and nothing of it: rendezvous (gloo), shard
    computation, barrier, MAX over ranks; rank 0 prints the shards.  Runs on a CPU-only box (tests/test_shard_gloo.py)."""
    import torch.distributed as dist
    from gama_tts_amd.shard import max_over_ranks

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    total, lo, hi = rank_workload(args, rank, world)
    spans = [None] * world
    if world > 1:
        dist.all_gather_object(spans, (lo, hi))
        dist.barrier()
        slowest = max_over_ranks(1.0 + rank, dist)
    else:
        spans = [(lo, hi)]
        slowest = 1.0
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "global_batch": total, "shards": [list(x) for x in spans],
                          "max_over_ranks_of_1_plus_rank": slowest, "launcher": "self" if os.environ.get("GVTM_BENCH_SELF_LAUNCHED") else "external"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


PARITY_TOL = {"f32": 0.0, "mixed": 1e-5, "f64": 1e-9}
PARITY_TOL_LONG_F64 = 5e-8  # 30 s in double: the device's and glibc's exp2 / pow differ in the last bit (tests/test_gpu_parity.py)


def parity_check(d_audio, host_tracks, rows, precision, delay, frames, output_rate, model=0):
    """Utterances `rows` of the TIMED output buffer against the oracle (peak-relative error; float: bit for bit)."""
    import numpy as np
    import oracle

    worst, same = 0.0, True
    for b in rows:
        got = d_audio[b].cpu().numpy()
        tr = host_tracks[b % len(host_tracks)][:frames]
        if model == 5:
            ref = oracle.synthesize5(oracle.male5_config(output_rate), tr)[0]
        else:
            ref = oracle.synthesize(oracle.male_config(output_rate, delay, 1 if model == 4 else 0, float_model=int(precision == "f32")), tr)
        if ref.size != got.size:
            return {"utterances": list(rows), "pass": False, "error": "sample count %d != oracle %d" % (got.size, ref.size)}
        same = same and bool(np.array_equal(got, ref))
        peak = float(np.abs(ref).max())
        if peak > 0:
            worst = max(worst, float(np.abs(got.astype(np.float64) - ref).max() / peak))
    tol = 2e-6 if model == 5 else (PARITY_TOL_LONG_F64 if (precision == "f64" and frames > 2000) else PARITY_TOL[precision])
    ok = same if precision == "f32" and model != 5 else worst <= max(tol, 6e-8)  # (double paths: a float32 sample may flip by one ulp)
    return {"utterances": list(rows), "max_err": worst, "bit_identical": same, "tolerance": "bit-identical" if (precision == "f32" and model != 5) else tol,
            "against": "oracle/vtm_oracle.c (pinned to reference-made vectors), same tracks, after the timed region", "pass": bool(ok)}


def end_to_end(g, plan, host_pool, batch, frames, n_timed, kernel_only_rate, label):
    """The host-buffer entries on page-locked buffers: H2D frames + kernel + D2H samples, float32 and int16 output
    (include/gama_vtm.h: gvtm_synthesize_batch_host, gvtm_synthesize_batch_host_pcm16)."""
    import numpy as np

    n_out = plan.output_count(frames)
    p_in = g.PinnedArray((batch, frames, 16), np.float32)
    reps = (batch + len(host_pool) - 1) // len(host_pool)
    for r in range(reps):
        lo = r * len(host_pool)
        n = min(len(host_pool), batch - lo)
        p_in.array[lo:lo + n] = host_pool[:n, :frames]
    raw = g.PinnedArray((batch * n_out,), np.float32)  # viewed as float32 [B][N] or int16 [B][N] (its first half)
    counts = np.zeros(batch, np.int64)
    out = []
    for kind in ("int16", "float32"):
        buf = raw.array.view(np.int16)[: batch * n_out].reshape(batch, n_out) if kind == "int16" else raw.array.reshape(batch, n_out)
        plan.synthesize_host_into(p_in.array, buf, None, counts, None)  # warm-up: staging buffers, streams
        plan.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(n_timed):
            plan.synthesize_host_into(p_in.array, buf, None, counts, None)
        el = (time.perf_counter() - t0) / n_timed
        kms, launches = plan.take_kernel_ms()  # the synthesis kernels of these calls (one per slice), HIP events
        plan.set_timing(False)
        kernels_ms = kms * launches / n_timed
        assert int(counts.min()) == n_out and int(counts.max()) == n_out
        rate = float(n_out) * batch / el
        out.append({"workload": label, "output": kind, "ms": el * 1e3, "value": rate, "unit": "samples/s",
                    "bytes_over_pcie": float(batch) * (frames * 64.0 + n_out * (2.0 if kind == "int16" else 4.0)),
                    "pcie_gbs": float(batch) * (frames * 64.0 + n_out * (2.0 if kind == "int16" else 4.0)) / el / 1e9,
                    "synthesis_kernels_ms": kernels_ms, "slices": launches // n_timed,
                    "vs_kernel_only": kernels_ms / (el * 1e3),
                    "kernel_only_value_of_this_workload": kernel_only_rate,
                    "host_buffers": "page-locked (gvtm_host_alloc = hipHostMalloc)"})
    p_in.close()
    raw.close()
    return out


def traffic_entry(batch, frames, delay, precision, model):
    """HBM bytes per launch of this workload from the committed rocprofv3 PMC passes (profiles/traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            key = "batch%d_frames%d_delay%d_%s%s" % (batch, frames, delay, precision,
                                                     "_model5" if model == 5 else ("_model4" if model == 4 else ""))
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves, BEFORE anything touches a GPU (this process never does)
        os.environ["GVTM_BENCH_SELF_LAUNCHED"] = "1"
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if args.rehearse_launch:
        return rehearse_launch(args)

    import numpy as np  # noqa: F401
    import torch

    import gama_tts_amd as g
    from gama_tts_amd import capi
    import tracks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d: the launcher started %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the VTM path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    model5 = args.model == 5
    model4 = args.model == 4
    global_batch, lo, hi = rank_workload(args, rank, world)
    batch = hi - lo
    voice = os.path.join(ROOT, "tests", "golden", "voice5_male.txt" if model5 else "voice_male.txt")
    cfgd = g.read_config_file(voice)
    PREC = {"f64": capi.PRECISION_F64, "mixed": capi.PRECISION_MIXED, "f32": capi.PRECISION_F32}

    def make_plan(precision, delay):
        if model5:
            return g.Plan(g.config5_from_dict(cfgd, args.output_rate), 250.0, local_rank)
        return g.Plan(g.config_from_dict(cfgd, args.output_rate, delay, PREC[precision], capi.TUBE_30_18 if model4 else capi.TUBE_10_6),
                      250.0, local_rank)

    plan = make_plan(args.precision, args.delay)
    n_out = plan.output_count(args.frames)
    steps_per_utt = args.frames * int(plan.info.control_steps)

    # synthetic tracks: SURVEY.md 8(d) config-2 generator, seeded by GLOBAL utterance id; a pool of distinct
    # tracks is tiled over the shard so that host-side generation stays cheap for big batches
    pool = max(1, min(batch, 256))
    host_pool = tracks.random_tracks(pool, args.frames, seed0=1000 + lo, consonant_heavy=model5)
    d_pool = torch.from_numpy(host_pool).to(dev)

    def tiled(n, frames):
        reps = (n + pool - 1) // pool
        return d_pool[:, :frames].repeat((reps, 1, 1))[:n].contiguous()

    d_params = tiled(batch, args.frames)
    d_audio = torch.empty((max(batch, 1), n_out), dtype=torch.float32, device=dev)
    d_counts = torch.zeros(max(batch, 1), dtype=torch.int64, device=dev)
    d_max = torch.zeros(max(batch, 1), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.synthesize_device(d_params, batch, args.frames, d_audio, n_out, None, d_counts, d_max, stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    plan.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = plan.take_kernel_ms()
    plan.set_timing(False)
    if batch:
        assert int(d_counts[:batch].min().item()) == n_out and int(d_counts[:batch].max().item()) == n_out
    parity = None
    if rank == 0 and batch and not args.no_parity_check:
        # two utterances of the buffer the timed launches wrote, against the oracle (outside the timed region)
        parity = parity_check(d_audio, host_pool, sorted({0, min(batch, pool) - 1}), args.precision, args.delay, args.frames, args.output_rate, args.model)
        if not parity["pass"]:
            raise SystemExit("bench.py: the timed output does not match the oracle: %s" % json.dumps(parity))

    from gama_tts_amd.shard import max_over_ranks
    elapsed = max_over_ranks(elapsed, dist, dev if args.dist_backend == "nccl" else None)

    total_samples = float(n_out) * global_batch * args.steps
    value = total_samples / elapsed
    algo_bytes = float(batch) * (args.frames * 64.0 + n_out * 4.0)
    algo_flops = float(batch) * (steps_per_utt * FLOP_PER_STEP + n_out * FLOP_PER_OUTPUT)
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
    tflops = algo_flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else None

    if rank == 0:
        tr = traffic_entry(batch, args.frames, args.delay, args.precision, args.model)
        ref_cls, ref_how = reference_class(args.precision, args.delay, args.model)
        line = {
            "metric": "audio samples/sec (whole node), batched VTM @%s" % ("48kHz" if model5 and args.output_rate == 48000.0 else "44.1kHz"),
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.global_batch is None else "strong",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "real_time_factor": value / args.output_rate,
            "config": {
                "workload": ("batch=%d/GPU x %d frames (%.1f s) synthetic parameter tracks, %s semantics, %s voice, %.0f Hz out%s" % (
                    batch, args.frames, args.frames * 0.004, ref_cls.split(" (")[0], "5_male" if model5 else "male", args.output_rate,
                    " = BASELINE configs[3] (4096 x 30 s, 2x oversampled tube)" if (batch, args.frames, args.delay, args.model) == (4096, 7500, 2, 0) else "")),
                "batch_per_gpu": batch,
                "global_batch": global_batch,
                "frames": args.frames,
                "section_delay": args.delay,
                "internal_rate_hz": float(plan.info.internal_rate_hz),
                "samples_per_utterance": n_out,
                "precision": args.precision,
                "reproduces": ref_cls,
                "parity": ref_how,
                "parallelism": "utterance-sharded x%d (contiguous shards), no collectives" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": tr.get("bytes") if tr else None,
                "traffic_source": ("replayed from profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (%s)"
                                   % ", ".join(tr.get("source", []))) if tr else None,
                "kernel": "gvtm::m5::vtm5_synth_kernel" if model5 else "gvtm::v2::vtm_synth_kernel",
                "kernel_ms": kernel_ms,
                "launches_timed": launches,
                "algorithmic_bytes_per_launch": algo_bytes,
                # the resource that binds: vector-ALU issue of lane-sparse serial code
                "valu": {
                    "achieved_tflops": tflops,
                    "peak": VALU_PEAK_TFLOPS[args.precision],
                    "unit": "TFLOP/s",
                    "frac": (tflops / VALU_PEAK_TFLOPS[args.precision]) if tflops else None,
                    "algorithmic_flops_per_launch": algo_flops,
                    "flop_per_output_sample": algo_flops / (float(batch) * n_out) if batch else None,
                    "valu_active": tr.get("valu_active") if tr else None,
                    "valu_active_source": tr.get("valu_active_source") if tr else None,
                },
            },
        }
        line["parity_check"] = parity
        plain = not model5 and not model4
        if world == 1 and not args.no_extras and plain:
            line["extras"] = extras(args, g, make_plan, tiled, d_audio, d_counts, stream, n_out, host_pool)
        if world == 1 and not args.no_end_to_end and plain:
            # kernel-only rates beside which the host-buffer entries are quoted
            k2s = next((e["value"] for e in line.get("extras", []) if (e["precision"], e["batch"], e["frames"], e["section_delay"]) == (args.precision, 4096, 500, 1)), None)
            e2e = end_to_end(g, make_plan(args.precision, 1), host_pool, 4096, 500, 3, k2s, "4096 x 500 frames (2 s), SectionDelay 1, %s" % args.precision)
            if (batch, args.frames) == (DEFAULT_BATCH_PER_GPU, DEFAULT_FRAMES):
                e2e += end_to_end(g, plan, host_pool, batch, args.frames, 2, value, "BASELINE configs[3]: %d x %d frames, SectionDelay %d, %s" % (batch, args.frames, args.delay, args.precision))
            line["end_to_end"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            # one core of this host through the compiled reference, once per reference class (float / double)
            by_class = {}
            for cls, fm in (("float", True), ("double", False)):
                if model5 and fm:
                    continue
                by_class[cls] = cpu_baseline(host_pool, args.output_rate, args.delay, float_model=fm, model5=model5, model4=model4)
            mine = "float" if (args.precision == "f32" and not model5) else "double"
            line["cpu_baseline"] = by_class[mine]
            line["cpu_baseline_by_class"] = by_class
            if plain and "extras" in line:
                d1 = {"float": cpu_baseline(host_pool[:, :500], args.output_rate, 1, budget_s=3.0, float_model=True),
                      "double": cpu_baseline(host_pool[:, :500], args.output_rate, 1, budget_s=3.0, float_model=False)}
                for e in line["extras"]:
                    ref = (by_class if e["section_delay"] == args.delay else d1)["float" if e["precision"] == "f32" else "double"]
                    e["cpu_baseline"] = {"value": ref["value"], "unit": "samples/s", "cores": 1, "kind": ref["kind"], "reference_class": ref["reference_class"],
                                         "gpu_over_one_core": e["value"] / ref["value"]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def extras(args, g, make_plan, tiled, d_audio, d_counts, stream, n_out_main, host_pool):
    """Other precisions and sizes on the same GPU, short runs beside the headline: the fp32 / fp64 sweep of
    configs[3], the batch BASELINE's target is quoted on with 2 s tracks, and configs[1] (256 x 2 s, VTM0).  The
    256 x 2 s entries also say how far each precision's samples are from the fp64 path's (which the tests hold
    within 1e-9 of the reference's double model): max |x - x64| / max |x64| over the batch."""
    import torch

    out = []
    flat = d_audio.view(-1)

    def run(precision, batch, frames, delay, n_timed, keep=None):
        pl = make_plan(precision, delay)
        n_out = pl.output_count(frames)
        steps = frames * int(pl.info.control_steps)
        dp = tiled(batch, frames)
        need = batch * n_out
        da = (flat[:need] if need <= flat.numel() else torch.empty(need, dtype=torch.float32, device=flat.device)).view(batch, n_out)
        dc = d_counts[:batch] if batch <= d_counts.numel() else torch.zeros(batch, dtype=torch.int64, device=flat.device)
        pl.synthesize_device(dp, batch, frames, da, n_out, None, dc, None, stream)
        torch.cuda.synchronize()
        pl.set_timing(True)
        t_e = time.perf_counter()
        for _ in range(n_timed):
            pl.synthesize_device(dp, batch, frames, da, n_out, None, dc, None, stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t_e
        kms, _ = pl.take_kernel_ms()
        assert int(dc.min().item()) == n_out
        par = None
        if not args.no_parity_check:
            par = parity_check(da, host_pool, sorted({0, min(batch, len(host_pool)) - 1}), precision, delay, frames, args.output_rate)
            if not par["pass"]:
                raise SystemExit("bench.py: the timed output of %s %d x %d does not match the oracle: %s" % (precision, batch, frames, json.dumps(par)))
        cls, _how = reference_class(precision, delay, 0)
        flops = float(batch) * (steps * FLOP_PER_STEP + n_out * FLOP_PER_OUTPUT)
        e = {"precision": precision, "batch": batch, "frames": frames, "section_delay": delay, "reproduces": cls,
             "value": float(n_out) * batch * n_timed / el, "unit": "samples/s", "kernel_ms": kms,
             "real_time_factor": float(n_out) * batch * n_timed / el / args.output_rate,
             "hbm_gbs": float(batch) * (frames * 64.0 + n_out * 4.0) / (kms * 1e-3) / 1e9,
             "valu_tflops": flops / (kms * 1e-3) / 1e12, "valu_frac": flops / (kms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS[precision],
             "parity_check": par}
        if keep is not None:
            keep[precision] = da.clone()
        del dp
        return e

    main_key = (args.precision, args.frames, args.delay)
    per_gpu = args.batch if args.batch is not None else DEFAULT_BATCH_PER_GPU
    # the sweep of configs[3] at the headline's size
    for p in ("mixed", "f64", "f32"):
        if (p, args.frames, args.delay) != main_key:
            out.append(run(p, per_gpu, args.frames, args.delay, 2 if args.frames > 2000 else 5))
    # 4096 x 2 s, VocalTractModel0 semantics (BASELINE's ">= 4096 utterances" target)
    for p in ("f32", "mixed", "f64"):
        out.append(run(p, 4096, 500, 1, 5))
    # configs[1]: 256 x 2 s, VocalTractModel0 semantics
    keep = {}
    small = [run(p, 256, 500, 1, 10, keep) for p in ("f64", "mixed", "f32")]
    ref = keep["f64"].double()
    peak = ref.abs().amax(dim=1).clamp_min(1e-300)
    for e in small:
        dev_ = ((keep[e["precision"]].double() - ref).abs().amax(dim=1) / peak)
        e["vs_f64_path"] = {"worst": float(dev_.max().item()), "median": float(dev_.median().item()),
                            "meets_1e-5": bool(dev_.max().item() <= 1e-5)}
    out.extend(small)
    return out


if __name__ == "__main__":
    main()
