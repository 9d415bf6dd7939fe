#!/usr/bin/env python3
"""Throughput of the batched VTM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (parameter frames in HBM -> audio samples in HBM) over
one batch of synthetic utterances.  N=1 runs BASELINE.json configs[1] (batch 256, 500 frames
= 2 s, "VTM0 fp32" = VocalTractModel0<float> semantics, 44.1 kHz out; the device output is
bit-identical to that reference class); the fp64 model and batch 4096 ride along under "extras".  For N>1 the driver starts one process per
GPU (torch.distributed.run); every rank synthesizes its own batch (weak scaling, independent
utterances, no data-path collective — SURVEY.md 8e); the only communication is the barrier
and the MAX of the elapsed time.

Rank 0 prints one JSON line.  `roofline.achieved` = algorithmic bytes per launch (64 B per
input frame + 4 B per output sample) / the synthesis kernel's mean duration
measured with HIP events on the launch stream.  `cpu_baseline` times the real reference
(oracle/_ref, compiled from /root/reference in the build container; "port" = our C oracle when
that binary is absent) on one host core over a bounded sample of the same workload.
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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(params, output_rate, delay, budget_s=12.0, float_model=False, model5=False, model4=False):
    """Single-thread CPU throughput (output samples/s) on a bounded sample of the workload."""
    import numpy as np
    import oracle
    voice = oracle.VOICE5_MALE if model5 else oracle.VOICE_MALE

    sample = params[:4]
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
        _, info = oracle.ref_synthesize(sample[0], model, output_rate, 250.0, config=voice, kind=exe_kind, repeat=2)
        per_rep = float(info["sec"]) / 2
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
        out = synth(sample[0])
        per = time.perf_counter() - t0
        repeat = max(1, int(budget_s / max(per, 1e-6) / len(sample)))
        total_samples = 0
        t0 = time.perf_counter()
        for tr in sample:
            for _ in range(repeat):
                total_samples += synth(tr).size
        total_sec = time.perf_counter() - t0
        desc = "%d utterances x %d repeats of %d frames through oracle/vtm_oracle.c" % (len(sample), repeat, sample.shape[1])
    return {"value": total_samples / total_sec, "unit": "samples/s", "cores": 1, "kind": kind, "sample": desc}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=500, help="control frames per utterance (4 ms each)")
    ap.add_argument("--delay", type=int, default=1, help="SectionDelay (1 = VocalTractModel0)")
    ap.add_argument("--precision", choices=["f64", "mixed", "f32"], default="f32",
                    help="f32 (default; BASELINE configs[1] is 'VTM0 fp32') = VocalTractModel0<float>, reference model 1, "
                         "output bit-identical to it; f64 = VocalTractModel0<double>, model 0 (reported under 'extras'); "
                         "mixed = fp64 with an fp32 resampler")
    ap.add_argument("--output-rate", type=float, default=None, help="default 44100 (48000 with --model 5, the 5_male voice's own rate)")
    ap.add_argument("--model", type=int, choices=[0, 4, 5], default=0,
                    help="0: the VocalTractModel0/2 path (default, BASELINE configs); 4: the 30+18-section tube of VocalTractModel4 "
                         "(--precision f64 = reference model 4, f32 = VocalTractModel4<float,1>; --delay ignored); 5: reference model 5 "
                         "(VocalTractModel5<double,1>, its own kernel; fp64, --delay / --precision ignored)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra single-GPU measurements (other precision, batch 4096)")
    ap.add_argument("--dist-backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    import numpy as np
    import torch

    import gama_tts_amd as g
    from gama_tts_amd import capi
    import tracks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
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
    if args.output_rate is None:
        args.output_rate = 48000.0 if model5 else 44100.0
    model4 = args.model == 4
    if model5:
        args.precision, args.delay = "f64", 1
    if model4:
        args.delay = 1
    voice = os.path.join(ROOT, "tests", "golden", "voice5_male.txt" if model5 else "voice_male.txt")
    cfgd = g.read_config_file(voice)
    prec = {"f64": capi.PRECISION_F64, "mixed": capi.PRECISION_MIXED, "f32": capi.PRECISION_F32}[args.precision]
    if model5:
        plan = g.Plan(g.config5_from_dict(cfgd, args.output_rate), 250.0, local_rank)
    else:
        plan = g.Plan(g.config_from_dict(cfgd, args.output_rate, args.delay, prec, capi.TUBE_30_18 if model4 else capi.TUBE_10_6), 250.0, local_rank)
    n_out = plan.output_count(args.frames)

    # synthetic tracks: SURVEY.md 8(d) config-2 generator; a pool of distinct tracks is tiled
    # over the batch so that host-side generation stays cheap for big batches
    pool = min(args.batch, 256)
    host_pool = tracks.random_tracks(pool, args.frames, seed0=1000 + 100000 * rank, consonant_heavy=model5)
    reps = (args.batch + pool - 1) // pool
    d_pool = torch.from_numpy(host_pool).to(dev)
    d_params = d_pool.repeat((reps, 1, 1))[: args.batch].contiguous()
    d_audio = torch.empty((args.batch, n_out), dtype=torch.float32, device=dev)
    d_counts = torch.zeros(args.batch, dtype=torch.int64, device=dev)
    d_max = torch.zeros(args.batch, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.synthesize_device(d_params, args.batch, args.frames, d_audio, n_out, None, d_counts, d_max, stream)

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
    assert int(d_counts.min().item()) == n_out and int(d_counts.max().item()) == n_out

    from gama_tts_amd.shard import max_over_ranks
    elapsed = max_over_ranks(elapsed, dist, dev if args.dist_backend == "nccl" else None)

    total_samples = float(n_out) * args.batch * world * args.steps
    value = total_samples / elapsed
    algo_bytes = float(args.batch) * (args.frames * 64.0 + n_out * 4.0)
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None

    # HBM traffic of this workload as measured with rocprofv3 PMC passes (profiles/traffic.json)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            key = "batch%d_frames%d_delay%d_%s%s" % (args.batch, args.frames, args.delay, args.precision,
                                                     "_model5" if model5 else ("_model4" if model4 else ""))
            traffic = json.load(f).get(key, {}).get("bytes")
    except (OSError, ValueError):
        pass

    if rank == 0:
        line = {
            "metric": "audio samples/sec (whole node), batched VTM @%s" % ("48kHz" if model5 and args.output_rate == 48000.0 else "44.1kHz"),
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == "f64" else "f32",
            "data": "synthetic",
            "real_time_factor": value / args.output_rate,
            "config": {
                "workload": ("batch=%d/GPU x %d frames (%.1f s) synthetic parameter tracks, VocalTractModel5<double> (reference "
                             "model 5), 5_male voice, %.0f Hz out" % (args.batch, args.frames, args.frames * 0.004, args.output_rate))
                            if model5 else
                            "batch=%d/GPU x %d frames (%.1f s) synthetic parameter tracks, VocalTractModel%s<%s> "
                            "semantics (SectionDelay %d), male voice, %.0f Hz out" % (
                                args.batch, args.frames, args.frames * 0.004, "4" if model4 else ("0" if args.delay == 1 else "2"),
                                "float" if args.precision == "f32" else "double", args.delay, args.output_rate),
                "batch_per_gpu": args.batch,
                "frames": args.frames,
                "samples_per_utterance": n_out,
                "precision": args.precision,
                "parallelism": "utterance-sharded x%d, no collectives" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic,
                "kernel": "gvtm::m5::vtm5_synth_kernel" if model5 else (
                    "gvtm::v2::vtm_synth_kernel" if os.environ.get("GVTM_KERNEL", "2") != "1" else "gvtm::v1::vtm_synth_kernel"),
                "kernel_ms": kernel_ms,
                "launches_timed": launches,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if world == 1 and not args.no_extras and not model5 and not model4:
            # the same kernel in the other arithmetic (BASELINE configs[1] names fp32, configs[3] an fp32/fp64
            # sweep) and at the batch BASELINE's target is quoted on; short runs, reported beside the headline
            def extra(precision, batch):
                pl = g.Plan(g.config_from_dict(cfgd, args.output_rate, args.delay,
                                               {"f64": capi.PRECISION_F64, "mixed": capi.PRECISION_MIXED, "f32": capi.PRECISION_F32}[precision]),
                            250.0, local_rank)
                reps_e = (batch + pool - 1) // pool
                dp = d_pool.repeat((reps_e, 1, 1))[:batch].contiguous()
                da = torch.empty((batch, n_out), dtype=torch.float32, device=dev)
                dc = torch.zeros(batch, dtype=torch.int64, device=dev)
                for _ in range(2):
                    pl.synthesize_device(dp, batch, args.frames, da, n_out, None, dc, None, stream)
                torch.cuda.synchronize()
                pl.set_timing(True)
                t_e = time.perf_counter()
                n_e = 5
                for _ in range(n_e):
                    pl.synthesize_device(dp, batch, args.frames, da, n_out, None, dc, None, stream)
                torch.cuda.synchronize()
                el = time.perf_counter() - t_e
                kms, _ = pl.take_kernel_ms()
                assert int(dc.min().item()) == n_out
                del da, dp
                return {"precision": precision, "batch": batch, "value": float(n_out) * batch * n_e / el, "unit": "samples/s",
                        "kernel_ms": kms, "real_time_factor": float(n_out) * batch * n_e / el / args.output_rate}
            other = "f32" if args.precision != "f32" else "f64"
            line["extras"] = [extra(other, args.batch), extra(args.precision, 4096), extra(other, 4096)]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_pool, args.output_rate, args.delay, float_model=args.precision == "f32", model5=model5, model4=model4)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
