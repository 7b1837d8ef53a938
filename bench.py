#!/usr/bin/env python3
"""Headline benchmark: concept heat maps per second on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One *step* = one generate_image-equivalent call of the hot path on synthetic inputs already
resident in HBM: flux-schnell geometry, 1024x1024 (4096 image tokens), 256 text tokens, 4 concepts,
4 sequential diffusion steps, heat maps of double blocks 15..18 in both spaces
(BASELINE.json configs[1]; random-init weights, seeded synthetic latents/embeddings).
N > 1: one process per GPU (torch.distributed.run), a full weight replica per GPU, work items
round-robin over ranks, ONE RCCL all_gather of the (C,64,64) fp32 maps at the end -- weak scaling.
Each GPU sends `--batch` (default 5) independent work items through every forward -- one launch per kernel for all
of them, so the last round of workgroups of every launch is full (5 x 17 GEMM row tiles, 5 x 408 attention
workgroups on 256 CUs) -- on `--streams` (default 1) HIP streams; per item the results are bit-identical to a
single-item call, and rank 0 re-runs the last timed item alone after the timed region to check exactly that
(`batched_equals_single`).  The LAST group of --batch items of every rank carries the per-launch HIP-event timing
behind `roofline.achieved`; the groups before it run unperturbed.  `--steps` should be a multiple of `--batch`
(a ragged last group is run and warmed up too, as a smaller forward).
Rank 0 prints ONE JSON line.  `value` = heat maps produced by all ranks / wall time (max over ranks).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

# ---- algorithmic FLOPs of the path: GEMM 2MNK, attention 4*Nq*Nk*D*heads,
#      concept attention counted for the C query rows only, heat map 2*C*L*H per space
def step_flops(p, L, T, C, double_blocks_only=False):
    """One DiT forward; double_blocks_only = the stop_after_multimodal_attentions forward of the encode / sweep paths."""
    H, MLP, NH, D = p.hidden_size, p.mlp_hidden, p.num_heads, p.head_dim
    lin_tok = 2 * H * 3 * H + 2 * H * H + 4 * H * MLP
    dbl = lin_tok * (L + T + C) + 4 * (L + T) ** 2 * D * NH + 4 * C * (C + L) * D * NH + 3 * 2 * H * 6 * H
    sgl = lin_tok * (L + T) + 4 * (L + T) ** 2 * D * NH + 2 * H * 3 * H
    io = 2 * L * p.in_channels * H * 2 + 2 * (T + C) * p.context_in_dim * H
    if double_blocks_only:
        return p.depth * dbl + io - 2 * L * p.in_channels * H
    return p.depth * dbl + p.depth_single_blocks * sgl + io


MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense block-scaled fp8 (v_mfma_scale_f32_16x16x128_f8f6f4), same table


def cpu_baseline(p, L, T, C, steps_per_call, n_layers_double, n_layers_single):
    """The CPU oracle (fp32 restatement of the reference, oracle/flux_oracle.py) timed on this
    host's cores on a bounded sample: one full-size double block + one single block, extrapolated
    to the 19+38 blocks x 4 steps of one call.  Reported next to the GPU number, never mixed in."""
    from conceptattention_amd.weights import synthetic_state_dict
    from oracle import flux_oracle as O
    from oracle.full_block_case import full_block_inputs
    # the GPU box grants a 1-GPU job a 16-core share of its host (more threads only oversubscribe)
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    case = full_block_inputs(p, T=T, C=C)
    sd = synthetic_state_dict(p, seed=0, prefix="double_blocks.0.")
    sd.update(synthetic_state_dict(p, seed=0, prefix="single_blocks.0."))
    rope_ti = O.rope_cos_sin(torch.cat((case["txt_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    rope_ci = O.rope_cos_sin(torch.cat((case["concept_ids"][0], case["img_ids"][0])), p.axes_dim, p.theta)
    x = torch.cat((case["txt"], case["img"]), 1)

    def dbl():
        return O.double_block(sd, "double_blocks.0.", p.num_heads, case["img"], case["txt"], case["vec"], rope_ti,
                              case["concepts"], case["concept_vec"], rope_ci)

    def sgl():
        return O.single_block(sd, "single_blocks.0.", p.num_heads, x, case["vec"], rope_ti)

    with torch.no_grad():
        dbl(), sgl()  # warm
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            dbl()
        td = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            sgl()
        ts = (time.perf_counter() - t0) / reps
    call_s = steps_per_call * (n_layers_double * td + n_layers_single * ts)
    return {"value": C / call_s, "unit": "concept-heatmaps/s", "cores": cores, "kind": "port",
            "sample": (f"fp32 oracle, 1 double block ({td:.2f} s) + 1 single block ({ts:.2f} s) at L={L} T={T} C={C}, "
                       f"{reps} reps each after 1 warm-up, extrapolated x({n_layers_double},{n_layers_single}) "
                       f"x{steps_per_call} steps = {call_s:.0f} s per call"),
            "seconds_per_call": call_s}


def stub_main(args):
    """--stub-workload: the N-rank plumbing of this file (init, item sharding, barrier-bracketed timing, the
    gather, max over ranks, rank 0's single JSON line) on CPU tensors; used by tests/test_bench_launcher_cpu.py."""
    from conceptattention_amd import distributed as D
    rank, world, _ = D.init_from_env()
    C = args.concepts
    n_timed = world * args.steps
    mine = D.shard_items(n_timed, rank, world)
    if args.stub_fail_rank is not None and rank == args.stub_fail_rank:
        raise SystemExit(f"stub: rank {rank} fails on request")
    D.barrier()
    t0 = time.perf_counter()
    local = torch.stack([torch.full((2, C, 4, 4), float(j)) for j in mine])
    if args.workload == "sweep":   # level-sharded table + ONE all_reduce(sum), as the real sweep workload
        allm = torch.zeros(n_timed, 2, C, 4, 4)
        allm[mine] = local
        D.allreduce_sum_(allm)
    else:                          # generate / encode: item-sharded, ONE all_gather
        allm = D.gather_heatmaps(local, n_timed, rank, world)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, "cpu")
    ok = all(bool((allm[j] == float(j)).all()) for j in range(n_timed))
    evidence = D.collective_evidence("cpu")   # the same record the real line carries as "collective"
    if args.stub_evidence:   # TEST ONLY: pretend the record came from an RCCL group with this many ranks / devices
        seen, distinct = (int(v) for v in args.stub_evidence.split(","))
        evidence = dict(evidence or {}, backend="rccl (torch.distributed 'nccl')", ranks_seen=seen, distinct_devices=distinct,
                        pci_bus_ids=[3 + (r if r < distinct else 0) for r in range(world)])   # (known bus ids: conclusive)
    D.check_collective_evidence(evidence, world)
    if rank == 0:
        print(json.dumps({"metric": "STUB (launcher test, no GPU work)", "value": n_timed * C / max(elapsed, 1e-9),
                          "workload": args.workload,
                          "collective": "all_reduce" if args.workload == "sweep" else "all_gather",
                          "collective_evidence": evidence,
                          "unit": "stub-items/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "gathered_in_item_order": ok,
                          "data": "stub"}), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def self_launch(n_gpus: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` without an outer launcher: start one fresh process per GPU through
    torch.distributed.run (rendezvous on 127.0.0.1) BEFORE this process has touched the GPU, let rank 0's JSON line
    go straight to our stdout and return the children's exit code.  (One process per device is also how the
    reference is used on several GPUs: experiments/imagenet_segmentation/run_experiment.py:56.)"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="flux-schnell")
    ap.add_argument("--workload", choices=("generate", "encode", "sweep"), default="generate",
                    help="generate: one generate_image-equivalent call per step (BASELINE.json configs[1]; with --model "
                         "flux-dev --concepts 8 --diffusion-steps 50: configs[2], one replica per GPU).  encode: one "
                         "encode_image-equivalent image per step (configs[3]: images sharded over the ranks, one "
                         "all_gather; default 2 concepts).  sweep: one noise level of the per-layer x noise-level table "
                         "per step (configs[4]: levels sharded over the ranks, one all_reduce(sum) of the table; add "
                         "--precision fp8 for the fp8 form)")
    ap.add_argument("--concepts", type=int, default=None, help="default 4 (2 for --workload encode)")
    ap.add_argument("--dump-maps", default=None,
                    help="rank 0 saves the gathered maps [items, 2, ...] to this .npy file (N-rank == 1-rank checks)")
    ap.add_argument("--diffusion-steps", type=int, default=4)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=5,
                    help="work items that share one forward (one launch per kernel for all of them): 5 x 17 row tiles "
                         "and 5 x 408 attention workgroups fill the 256 CUs to 99.6 %% in the last round of every "
                         "launch; 1 = one item per forward")
    ap.add_argument("--streams", type=int, default=1,
                    help="groups of --batch items kept in flight per GPU on separate HIP streams")
    ap.add_argument("--precision", choices=("bf16", "fp8"), default="bf16",
                    help="fp8: the large projections run on e4m3 operands (reduced-precision mode of "
                         "BASELINE.json configs[4]; NOT the headline metric, which is bf16)")
    ap.add_argument("--residual", choices=("fp32", "bf16"), default="fp32",
                    help="storage type of the residual streams (fp32 = the product default; bf16 = the reference's "
                         "activation dtype, an A/B aid: see DESIGN.md section 2 for what it costs in heat-map error)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the per-launch HIP-event timing of the GEMM kernel (roofline.achieved then "
                         "comes from whole-path FLOPs / wall time)")
    ap.add_argument("--no-solo-check", action="store_true",
                    help="skip the single-item re-run of the last timed item after the timed region "
                         "(batched_equals_single is then null); profiling runs use it so that every launch of the "
                         "process has the batched shape")
    ap.add_argument("--no-block-timing", action="store_true",
                    help="skip the extra, untimed forward that times the 19 double blocks (concept_attention_block)")
    ap.add_argument("--profile-mode", action="store_true",
                    help="what tools/profile_round.py runs under rocprofv3: --no-solo-check --no-block-timing "
                         "--no-cpu-baseline --no-kernel-timing, so the process holds exactly 1 warm-up group + the timed "
                         "groups of --batch items and every launch in the trace has the timed shape; launches are counted "
                         "per kernel kind (no events) and reported as launch_counts")
    ap.add_argument("--stub-fail-rank", type=int, default=None,
                    help="TEST ONLY (with --stub-workload): this rank exits non-zero after the rendezvous")
    ap.add_argument("--stub-evidence", default=None,
                    help="TEST ONLY (with --stub-workload): 'ranks_seen,distinct_devices' reported as if by an RCCL group")
    ap.add_argument("--stub-workload", action="store_true",
                    help="TEST ONLY: replace the GPU work of an item by a constant CPU tensor so that the launcher, "
                         "sharding, gather and JSON plumbing can be exercised without a GPU (gloo); the line says so")
    args = ap.parse_args()
    if args.concepts is None:
        args.concepts = 2 if args.workload == "encode" else 4
    if args.profile_mode:
        args.no_solo_check = args.no_block_timing = args.no_cpu_baseline = args.no_kernel_timing = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:  # before any rendezvous is attempted
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: run "
                         f"`python bench.py --gpus {args.gpus}` (self-launching) or torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if args.stub_workload:
        return stub_main(args)

    from conceptattention_amd import distributed as D
    from conceptattention_amd import ops
    from conceptattention_amd import _lib as L
    from conceptattention_amd.params import T5_TOKENS, configs
    from conceptattention_amd.pipeline import ConceptAttentionFluxPipeline
    from conceptattention_amd.weights import synthetic_inputs

    rank, world, local = D.init_from_env()
    # CA_BENCH_DEVICE pins every rank to one device (multi-process rehearsal on a 1-GPU box, with
    # CA_DIST_BACKEND=gloo since RCCL refuses two ranks on one GPU); normally rank r uses GPU LOCAL_RANK
    dev = torch.device(os.environ.get("CA_BENCH_DEVICE") or f"cuda:{local}")
    torch.cuda.set_device(dev)
    if L.load().ca_check_device() != 0:
        raise SystemExit(L.load().ca_last_error().decode())

    p = configs[args.model]
    C, T = args.concepts, T5_TOKENS[args.model]
    Lp = (args.size // 16) ** 2
    pipe = ConceptAttentionFluxPipeline(args.model, device=dev, weights="synthetic", weight_seed=0,
                                        precision=args.precision,
                                        residual_dtype=torch.float32 if args.residual == "fp32" else torch.bfloat16)
    layer_indices = list(range(15, 19))
    n_timed = world * args.steps  # timed work items 0..n_timed-1, item i on rank i % world

    # every work item's inputs are generated and made resident in HBM before timing
    def item_inputs(j):
        inp = synthetic_inputs(p, args.size, args.size, T, C, seed=1000 + j, device="cpu", dtype=torch.bfloat16)
        return {k: inp[k].to(dev) for k in ("latent", "txt", "vec", "concepts")}

    timed_items = D.shard_items(n_timed, rank, world)
    # warm-up: W steps asked; a full group of --batch items is run whenever W > 0, so that the activation set and
    # every kernel of the batched shape exist before the timed region (more warm-up than asked, never less)
    n_warm = 0 if args.warmup <= 0 else max(args.warmup, args.batch)
    warm_items = [n_timed + rank * n_warm + i for i in range(n_warm)]
    inputs = {j: item_inputs(j) for j in timed_items + warm_items}

    wl = args.workload
    SWEEP_STEPS = 50   # the schedule the noise levels index (test_segmentations_per_time.py:75-104 sweeps 50 levels)
    if wl == "generate":
        gen_kw = dict(layer_indices=layer_indices, num_inference_steps=args.diffusion_steps, guidance=0.0)

        def run_many(js, n_streams, batch):
            res = pipe.generate_many_on_device([inputs[j] for j in js], n_streams=n_streams, batch=batch, **gen_kw)
            return [torch.stack((hm[0], cm[0])) for _, hm, cm in res]           # [2, C, side, side] per item
    elif wl == "encode":
        enc_kw = dict(layer_indices=layer_indices, num_samples=1, num_steps=4, noise_timestep=2, seed=0)

        def run_many(js, n_streams, batch):
            res = pipe.encode_many_on_device([inputs[j] for j in js], n_streams=n_streams, batch=batch, **enc_kw)
            return [torch.stack((ho[0], hc[0])) for ho, hc in res]
    else:   # sweep: every item is one noise level of the SAME image (item 0's inputs); all 19 layers per level
        img0 = item_inputs(0)

        def run_many(js, n_streams, batch):
            out, cross = pipe.layer_noise_sweep_on_device(img0["latent"], img0["txt"], img0["vec"], img0["concepts"],
                                                          [j % SWEEP_STEPS for j in js], num_steps=SWEEP_STEPS,
                                                          batch=batch)
            return [torch.stack((out[k], cross[k])) for k in range(len(js))]   # [2, 19, C, side, side] per level

    def run_item(j):
        return run_many([j], 1, 1)[0]

    if warm_items:
        # (encode / sweep: a group is one ~0.1 s forward; the chip needs a few of them after the idle seconds of weight
        # and input set-up before its clocks are where a sustained run holds them -- a timed region that starts on the
        # second forward after idle measured 110-150 ms per forward for the same 106 ms of kernels, round 5)
        # (--profile-mode keeps exactly one warm-up group: tools/profile_round.py checks Calls == 2 x the counted launches)
        for _ in range(1 if (wl == "generate" or args.profile_mode) else 5):
            run_many(warm_items, args.streams, args.batch)
        # a ragged group (steps not a multiple of --batch) is a forward of another shape: build its activation set
        # and kernels before the timed region as well (HipFluxDiT keeps the few most recent activation sets)
        for n_r in sorted({len(timed_items) % args.batch, len(timed_items[:-min(args.batch, len(timed_items))])
                           % args.batch} - {0}):
            run_many(warm_items[:n_r], 1, args.batch)

    # ---- per-launch timing of the GEMM kernel with HIP events on the launch stream.  Every event pair
    # costs a ~5 us pipeline drain around the launch (616 GEMM launches per call = 2.3 % of a call), so
    # the events are recorded during the LAST timed step only; the other timed steps run unperturbed.
    records = []
    EPI_NAMES = {0: "bias", 1: "gelu", 2: "gate*x+residual", 3: "split(bias|gelu)", 4: "qk-norm+rope(+gelu)"}

    def has_main_tiles(arr):
        """Does this call launch the tile kernel at all?  A problem of <= 128 rows (the modulation GEMM's 40 vectors) is
        one thin last row tile and runs in ca_gemm_thin_kernel only."""
        return any(arr[i].M > 256 or not (0 < arr[i].M % 256 <= 128) for i in range(len(arr)))

    def hook(arr, tile, launch):
        if not has_main_tiles(arr):     # not a launch of the kernel the roofline is about
            return launch()
        fl = sum(2.0 * arr[i].M * arr[i].N * arr[i].K for i in range(len(arr)))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        shape = (f"M{'+'.join(str(arr[i].M) for i in range(len(arr)))} N{arr[0].N} K{arr[0].K} "
                 f"{EPI_NAMES.get(arr[0].epilogue, arr[0].epilogue)}")
        records.append((tile, fl, s, e, shape))

    attn_records = []

    def attn_hook(arr, num_heads, launch):
        # algorithmic FLOPs of the launch: 4 * Nq * Nk * 128 * heads per problem
        fl = sum(4.0 * arr[i].nq * (arr[i].n0 + arr[i].n1) * 128 * num_heads for i in range(len(arr)))
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        attn_records.append((fl, s, e))

    torch.cuda.synchronize()
    D.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()   # the same region on the GPU's clock (stream time from first to last launch), reported beside the wall time
    local_maps = []
    # the LAST group of --batch items runs with the per-launch HIP-event timing; the groups before it unperturbed
    n_last = min(max(1, args.batch), len(timed_items))
    head, last = timed_items[:-n_last], timed_items[-n_last:]
    if head:
        local_maps += run_many(head, args.streams, args.batch)
    launch_counts = {}

    def count_gemm(arr, tile, launch):   # --profile-mode: no events, only how many calls of which tile kind were made
        key = f"gemm_tile_{tile}" if has_main_tiles(arr) else "gemm_thin_rows_only"
        launch_counts[key] = launch_counts.get(key, 0) + 1
        launch()

    def count_attn(arr, num_heads, launch):
        launch_counts["attn"] = launch_counts.get("attn", 0) + 1
        launch()

    if not args.no_kernel_timing:
        ops.set_gemm_hook(hook)
        ops.set_attn_hook(attn_hook)
    elif args.profile_mode:
        ops.set_gemm_hook(count_gemm)
        ops.set_attn_hook(count_attn)
    local_maps += run_many(last, 1, args.batch)
    ops.set_gemm_hook(None)
    ops.set_attn_hook(None)
    local_maps = torch.stack(local_maps)
    if wl == "sweep":
        # level-sharded table: every rank fills the rows of its own levels, ONE all_reduce(sum)
        all_maps = torch.zeros((n_timed,) + tuple(local_maps.shape[1:]), device=dev)
        all_maps[timed_items] = local_maps
        D.allreduce_sum_(all_maps)
    else:
        # the one collective of the job: gather the small fp32 maps of all ranks in item order (RCCL over xGMI)
        all_maps = D.gather_heatmaps(local_maps, n_timed, rank, world)
    ev1.record()
    t_enqueued = time.perf_counter() - t0
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    gpu_elapsed = ev0.elapsed_time(ev1) * 1e-3
    elapsed = D.max_over_ranks(elapsed, dev)

    calls = n_timed
    # evidence that the job's collectives ran over `world` ranks on `world` different devices (rank 0 prints it)
    collective = D.collective_evidence(dev) if world > 1 else None
    # ... and a group that is not what --gpus asked for ends the run non-zero instead of printing a line (every rank
    # holds the same record, so every rank exits)
    D.check_collective_evidence(collective, world, rehearsal=bool(os.environ.get("CA_BENCH_DEVICE")))

    # ---- the "concept-attention block" figure of BASELINE.json: average duration of a double block
    # (HIP events around each of the 19 block calls of one extra, untimed forward; rank 0 only)
    block = None
    if rank == 0 and not args.no_block_timing:
        m = pipe.model
        nb = max(1, min(args.batch, len(timed_items)))   # the forward holds --batch items; times are per item
        i0 = {k: torch.cat([inputs[j][k] for j in timed_items[:nb]], 0) for k in ("latent", "txt", "vec", "concepts")}
        from conceptattention_amd import sampling as S_
        con, con_ids, con_vec = S_.concept_inputs(i0["concepts"], i0["vec"])
        inp0 = S_.prepare_from_embeddings(i0["latent"].to(dev, torch.bfloat16), i0["txt"], i0["vec"])
        evs = []
        orig = m._double_block

        def timed_block(*a, **k):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            orig(*a, **k)
            e_.record()
            evs.append((s_, e_))
        m._double_block = timed_block
        try:
            m(img=inp0["img"], img_ids=inp0["img_ids"], txt=inp0["txt"], txt_ids=inp0["txt_ids"], concepts=con,
              concept_ids=con_ids, concept_vec=con_vec, y=inp0["vec"], timesteps=torch.full((nb,), 1.0, device=dev),
              guidance=torch.zeros(nb, device=dev), stop_after_multimodal_attentions=True, return_vectors=False)
        finally:
            m._double_block = orig
        torch.cuda.synchronize()
        us = sorted(s_.elapsed_time(e_) * 1e3 / nb for s_, e_ in evs)
        med = us[len(us) // 2]
        H_, MLP_, NH_, D_ = p.hidden_size, p.mlp_hidden, p.num_heads, p.head_dim
        dbl_flops = ((2 * H_ * 3 * H_ + 2 * H_ * H_ + 4 * H_ * MLP_) * (Lp + T + C) + 4 * (Lp + T) ** 2 * D_ * NH_
                     + 4 * C * (C + Lp) * D_ * NH_ + 3 * 2 * H_ * 6 * H_)
        block = {"what": f"ModifiedDoubleStreamBlock-equivalent (7 launches for {nb} work items), median of 19, "
                         "per work item", "us": med,
                 "tflop": dbl_flops / 1e12, "achieved_tflops": dbl_flops / med / 1e6,
                 "mfma_frac": dbl_flops / med / 1e6 / MFMA_BF16_PEAK_TFLOPS}
    if not bool(torch.isfinite(all_maps).all().item()):
        raise SystemExit("bench: non-finite heat maps")
    # the softmax over the concepts makes every patch's maps sum to 1 (concept_attention_pipeline.py:64-65); cheap, and
    # it catches a forward that produced finite garbage
    col = all_maps.float().sum(dim=-3)
    if float((col - 1.0).abs().max().item()) > 1e-3:
        raise SystemExit(f"bench: heat maps are not normalised over the concepts (max |sum - 1| = "
                         f"{float((col - 1.0).abs().max().item()):.3e})")
    # ---- the batched forward against the reference's unit of work (ONE item per call,
    # concept_attention_pipeline.py:115-202): the last timed item of rank 0 again, alone, outside the timed region;
    # its maps must equal the ones the batched group produced bit for bit
    batched_equals_single = None
    if rank == 0 and timed_items and not args.no_solo_check:
        solo = run_item(timed_items[-1])
        torch.cuda.synchronize()
        batched_equals_single = bool(torch.equal(solo, local_maps[-1]))
        if not batched_equals_single and args.precision == "bf16":
            # per item the batched forward is specified to be bit-identical to a single-item call (same MFMA, same
            # k order); a line whose value comes from a forward that is not must not exist
            raise SystemExit("bench: the batched forward differs from the single-item call of the same item "
                             f"(max |diff| = {float((solo - local_maps[-1]).abs().max().item()):.3e})")

    if rank == 0:
        if wl == "generate":
            flops_call = args.diffusion_steps * step_flops(p, Lp, T, C) + \
                len(layer_indices) * args.diffusion_steps * 2 * (2 * C * Lp * p.hidden_size)
            maps_per_item = C
            metric = f"concept-heatmaps/sec ({args.size}x{args.size}, {C} concepts, {args.diffusion_steps} steps)"
            workload = (f"{args.model} {args.precision} {args.size}x{args.size}, {C} concepts, "
                        f"{args.diffusion_steps} diffusion steps, {T} text tokens "
                        "(generate_image-equivalent call; random-init weights, synthetic latents/embeddings)")
        elif wl == "encode":
            flops_call = step_flops(p, Lp, T, C, True) + len(layer_indices) * 2 * (2 * C * Lp * p.hidden_size)
            maps_per_item = C
            metric = (f"concept-heatmaps/sec (encode_image path: {args.size}x{args.size}, {C} concepts, one forward of "
                      "the 19 double blocks per image)")
            workload = (f"{args.model} {args.precision} {args.size}x{args.size}, {C} concepts, {T} text tokens, "
                        "encode_image-equivalent call per image (add_noise_to_image at schedule[2] of 4 + ONE "
                        "stop_after_multimodal_attentions forward; random-init weights, synthetic latents/embeddings); "
                        "images sharded over the ranks, one all_gather")
        else:
            flops_call = step_flops(p, Lp, T, C, True) + p.depth * 2 * (2 * C * Lp * p.hidden_size)
            maps_per_item = C * p.depth
            metric = (f"concept-heatmaps/sec (per-layer x noise-level sweep: {args.size}x{args.size}, {C} concepts x "
                      f"{p.depth} double blocks per noise level)")
            workload = (f"{args.model} {args.precision} {args.size}x{args.size}, {C} concepts, {T} text tokens, one noise "
                        f"level of {SWEEP_STEPS} per step (one stop_after_multimodal_attentions forward, maps of all "
                        f"{p.depth} double blocks in both spaces); levels sharded over the ranks, one all_reduce(sum)")
        path_tflops = flops_call * calls / elapsed / 1e12 / world  # per GPU
        fp8 = args.precision == "fp8"
        roof = {"bound": "mfma", "peak": MFMA_FP8_PEAK_TFLOPS if fp8 else MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "traffic": None}
        if records:
            torch.cuda.synchronize()
            by_tile = {}
            by_shape = {}
            for tile, fl, s, e, shape in records:
                d = by_tile.setdefault(tile, [0.0, 0.0, 0])
                d[0] += fl
                d[1] += s.elapsed_time(e) * 1e-3
                d[2] += 1
                b = by_shape.setdefault((tile, shape), [0.0, 0.0, 0])
                b[0] += fl
                b[1] += s.elapsed_time(e) * 1e-3
                b[2] += 1
            tile, (fl, sec, n) = max(by_tile.items(), key=lambda kv: kv[1][1])
            # the same figure per launch shape of the dominant kernel (in situ: the model's own operands and epilogues)
            shapes = {sh: {"launches": b[2], "avg_launch_us": b[1] / b[2] * 1e6, "achieved": b[0] / b[1] / 1e12,
                           "share_of_kernel_time": b[1] / sec}
                      for (tl, sh), b in sorted(by_shape.items(), key=lambda kv: -kv[1][1]) if tl == tile and b[1] > 0.005 * sec}
            names = {1: "ca_gemm_kernel<8,4> (256x256x64)", 2: "ca_gemm_kernel<8,3> (256x192x64)",
                     3: "ca_gemm_kernel<8,2> (256x128x64)", 4: "ca_gemm_kernel<8,1> (256x64x64)",
                     5: "ca_gemm_pp_kernel<2,2> (256x256x64 ping-pong)", 6: "ca_gemm_pp_kernel<1,1> (256x128x64 ping-pong)",
                     7: "ca_gemm_pp_kernel<2,1> (256x192x64 ping-pong)"}
            if fp8:
                names[5] = "ca_gemm_pp_kernel<2,2,fp8> (256x256x128 ping-pong, e4m3)"
            roof.update(kernel=names.get(tile, str(tile)), launches=n, avg_launch_us=sec / n * 1e6,
                        flops_per_launch=fl / n, achieved=fl / sec / 1e12,
                        flops_are="executed 2*M*N*K of every problem of the call, which includes the low-plane q "
                                  "projection of the captured layers (about 0.55 % above the algorithmic count)",
                        recompute_from_csv="profiles/rNN_rocprofv3_kernel_stats.csv: avg_launch_us ~= (TotalDurationNs["
                                           "ca_gemm_pp_kernel<2,2>] + TotalDurationNs[ca_gemm_thin_kernel<2>]) / Calls["
                                           "ca_gemm_pp_kernel<2,2>] / 1e3 (HIP events add the ~5 us drain per call)",
                        timed_on=f"last group of {n_last} work items of rank 0 (one forward per diffusion step)",
                        launch_is="one ca_gemm_bf16 call: the ping-pong launch plus, for a thin last row tile, "
                                  "its thin-row launch (ca_gemm_thin_kernel); rocprofv3 lists the two kernels separately",
                        share_of_that_group=sec / (elapsed * n_last / max(len(timed_items), 1)),
                        by_shape=shapes)
        else:
            roof.update(kernel="whole path", achieved=path_tflops)
        roof_attn = None
        if attn_records:   # second MFMA kernel of the path, timed the same way on the same solo step
            fl = sum(r[0] for r in attn_records)
            sec = sum(r[1].elapsed_time(r[2]) for r in attn_records) * 1e-3
            attn_kernel = "ca_attn4_kernel (one wave per SIMD, 4 x 64 query rows x 64-key tiles, generated stream)"
            roof_attn = {"bound": "mfma", "kernel": attn_kernel,
                         "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "launches": len(attn_records),
                         "avg_launch_us": sec / len(attn_records) * 1e6, "flops_per_launch": fl / len(attn_records),
                         "achieved": fl / sec / 1e12, "frac": fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                         "share_of_that_group": sec / (elapsed * n_last / max(len(timed_items), 1)),
                         "timed_on": f"last group of {n_last} work items of rank 0", "traffic": None}
        # HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside the process,
        # so this is the rocprofv3 FETCH_SIZE/WRITE_SIZE measurement of this same command, committed
        # under profiles/ (method and the gfx950 x2 FETCH_SIZE correction are recorded in the file)
        try:
            import glob
            # a per-launch average is only this launch's traffic if every profiled launch had the timed shape: every
            # file states its launch mix (tools/profile_round.py); the newest one of THIS run's mix is used, none otherwise
            pmc_doc = pmc_file = None
            seen = []
            for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
                doc = json.load(open(cand))
                mix = doc.get("launch_mix") or {}
                seen.append(os.path.basename(cand))
                if (mix.get("items_per_forward") == args.batch and mix.get("workload") == wl
                        and mix.get("model") == args.model and mix.get("concepts") == C
                        and mix.get("size") == args.size and mix.get("only_batched_launches") is True
                        and mix.get("precision", "bf16") == args.precision):
                    pmc_doc, pmc_file = doc, cand
                    break
            if pmc_doc is None:
                roof["traffic_source"] = (f"no profiles/r*_pmc_hbm_traffic.json holds this run's launch mix (looked at "
                                          f"{seen[:4]}): traffic not reported")
            else:
                pmc = pmc_doc["kernels"]
                key = roof.get("kernel", "").split(" ")[0]
                if key in pmc and not fp8:
                    roof["traffic"] = pmc[key]["bytes_per_launch"]
                    roof["traffic_source"] = (f"profiles/{os.path.basename(pmc_file)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                              f"passes of `{pmc_doc.get('command', 'bench.py --profile-mode')}` at "
                                              f"{pmc_doc.get('git_head', 'an earlier commit')}: every launch of that process is a "
                                              f"{args.batch}-item launch; counters cannot be read from inside the process)")
                akey = roof_attn["kernel"].split(" ")[0] if roof_attn is not None else None
                if akey in pmc and not fp8:
                    roof_attn["traffic"] = pmc[akey]["bytes_per_launch"]
        except (OSError, KeyError, ValueError, IndexError):
            pass
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["path_achieved"] = path_tflops
        roof["path_frac"] = path_tflops / MFMA_BF16_PEAK_TFLOPS
        if fp8:
            roof["note"] = ("path_frac and concept_attention_block.mfma_frac are fractions of the 2.5 PFLOP/s bf16 "
                            "peak (attention and the small kernels stay bf16); frac is the e4m3 GEMM kernel "
                            "against the 5 PFLOP/s fp8 peak")
        # what exactly runs on e4m3 operands in fp8 mode, per workload (the sweep's line did not say that the
        # qkv projection of all 19 captured layers stays bf16)
        if wl == "generate":
            fp8_scope = (f"qkv / proj / mlp.0 / mlp.2 / linear1 / linear2 of every block EXCEPT double blocks {layer_indices}, "
                         "which stay bf16 entirely (the layers whose maps are returned; pipeline.fp8_keep_heatmap_layers): "
                         f"{p.depth - len(layer_indices)} of {p.depth} double blocks and all {p.depth_single_blocks} single blocks in e4m3")
        else:
            cap = layer_indices if wl == "encode" else list(range(p.depth))
            fp8_scope = (f"proj / mlp.0 / mlp.2 of all {p.depth} double blocks and the qkv projection of the "
                         f"{p.depth - len(cap)} double blocks whose maps are not requested in e4m3; the qkv projection (and the "
                         f"low-plane q projection) of the {len(cap)} captured layers stays bf16 "
                         "(HipFluxDiT.fp8_bf16_qkv_when_captured: their q / k / v ARE the vectors of the maps) -- "
                         f"{75 + 25 * (p.depth - len(cap)) // p.depth} % of the double blocks' projection FLOPs in e4m3; attention, "
                         "LayerNorm, heat maps bf16 / fp32 as always")
        res = {
            "metric": metric,
            "value": calls * maps_per_item / elapsed,
            "unit": "concept-heatmaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8(e4m3 projections)+bf16" if fp8 else "bf16", "data": "synthetic",
            "config": {"workload": workload,
                       "calls_per_step_per_gpu": 1,
                       "heatmap_layers": layer_indices if wl != "sweep" else list(range(p.depth)),
                       "tflop_per_call": flops_call / 1e12, "parallelism": f"replica x{world} (work items round-robin)",
                       "items_per_forward": args.batch, "streams_per_gpu": args.streams,
                       "residual_stream": args.residual,
                       **({"fp8_scope": fp8_scope} if fp8 else {})},
            "calls_per_s": calls / elapsed,
            "timed_region": {"wall_s": elapsed, "gpu_stream_s": gpu_elapsed, "host_enqueue_s": t_enqueued,
                             "note": "rank 0: wall = max over ranks of barrier-to-barrier time (what value uses); "
                                     "gpu_stream = HIP events on the launch stream around the same region; "
                                     "host_enqueue = host time until the last launch was queued"},
            "batched_equals_single": batched_equals_single,
            **({"collective": collective} if collective is not None else {}),
            **({"launch_counts": launch_counts} if args.profile_mode else {}),
            "roofline": roof,
            "roofline_attention": roof_attn,
            "concept_attention_block": block,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(p, Lp, T, C, args.diffusion_steps if wl == "generate" else 1, p.depth,
                                               p.depth_single_blocks if wl == "generate" else 0)
            if wl == "sweep":   # per level the oracle produces C x depth maps too
                res["cpu_baseline"]["value"] *= p.depth
        if args.dump_maps:
            import numpy as np
            np.save(args.dump_maps, all_maps.float().cpu().numpy())
        print(json.dumps(res), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
