#!/usr/bin/env python
"""bench.py - closures/s of the device-resident multi-view SMPL fit on MI355X.

Metric (BASELINE.json): L-BFGS closure evaluations per second (forward + backward, summed over all
concurrently fitted problems) for 8-view, 1-person problems; ms to convergence per frame; the LBS
vertex pass as a fraction of the HBM roofline.

A "step" = ONE complete 4-stage fit (reference cfg_files/fit_smpl.yaml weights, L-BFGS lr=1,
max_iter=30, history=100, strong-Wolfe; outer maxiters=30, ftol=gtol=1e-9) of this rank's batch of
32 synthetic frames x 8 views x 1 person (BASELINE configs[1]), every closure with its full
6890-vertex LBS pass like the reference's return_verts=True.  Inputs are resident in HBM before the
timed region.  Ranks fit disjoint frames (weak scaling: 32 frames per GPU; --strong: the 32 frames
split over the ranks); the only collective is the final all_gather of the fitted parameters (RCCL).

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --config demo        # BASELINE configs[0]: the reference's shipped demo inputs (1 frame, 6 views, VPoser)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mvsmplfitting_amd import _lib                      # noqa: E402
from mvsmplfitting_amd import synthetic as syn          # noqa: E402
from mvsmplfitting_amd.engine import MvFit, stage_weights   # noqa: E402
from mvsmplfitting_amd.sharding import gather_results, shard_range   # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CONST_BYTES = 82680 + 826800 + 17114760 + 661440      # v_template + shapedirs + posedirs + lbs_weights
PER_PROBLEM_BYTES = 2032 + 82680                      # (betas, pose_feature, A, transl) in + vertices out
PMC_JSON = os.path.join(ROOT, 'profiles', 'r2_pmc.json')     # written by tools/pmc_vertex_pass.py from rocprofv3 --pmc passes
# the imported reference itself (PyTorch CPU, create_fitting_closure + LBFGSLs + run_fitting) timed in the survey's
# build container (it cannot travel to the GPU box): closures/s inside L-BFGS, 8 vCPUs
SURVEY_REFERENCE_CLOSURES_PER_S = (74.0, 125.0)


def bytes_fwd(B, skin_topk=0):
    """Algorithmic bytes of one LBS vertex pass over B problems; with
    k-sparse skinning weights the weight matrix is k (weight, joint) pairs per vertex instead of 24 floats."""
    const = CONST_BYTES if not skin_topk else CONST_BYTES - 661440 + 6890 * skin_topk * 8
    return const + PER_PROBLEM_BYTES * B


def pmc_value(key, kernel_substr, field='traffic_bytes'):
    """A per-launch figure of the committed PMC summary (profiles/r2_pmc.json), or None."""
    try:
        with open(PMC_JSON) as f:
            d = json.load(f)
        for k, e in d.get(key, {}).items():
            if kernel_substr in k and field in e:
                return round(e[field])
    except Exception:
        pass
    return None


def pmc_traffic(key, kernel_substr):
    """Measured HBM-side bytes per launch, or None."""
    return pmc_value(key, kernel_substr, 'traffic_bytes')


def mfma_util(key, kernel_substr, launch_us):
    """Matrix-pipe utilisation of a launch: SQ_VALU_MFMA_BUSY_CYCLES (PMC, summed over the chip's SIMDs) over launch
    duration x 2.4 GHz x 1024 SIMDs.  Small by design: the contraction is the only MFMA work of the path (north_star)."""
    busy = pmc_value(key, kernel_substr, 'mfma_busy_cycles_per_launch')
    return None if busy is None else round(busy / (launch_us * 1e-6 * 2.4e9 * 1024), 4)


def build_inputs(eng, frames, views, seed0):
    """Synthetic config-2 inputs: GT parameter draws -> keypoints (by the GPU forward) -> noisy 2-D
    observations + confidences; initial parameters = zeros, scale 1."""
    cams = syn.make_camera_ring(views)
    fr = syn.make_frames(frames, seed0=seed0)
    xgt = np.zeros((frames, 118), np.float32)
    for k, (a, b) in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85),
                          scale=(85, 86)).items():
        xgt[:, a:b] = fr[k]
    eng.set_problems(cams, np.zeros((frames, views, 17, 2), np.float32), np.ones((frames, views, 17), np.float32))
    _, joints = eng.vertices(xgt)
    gt, conf = syn.make_observations(joints.cpu().numpy(), cams, seed=seed0 + 7)
    eng.set_problems(cams, gt, conf)
    x0 = np.zeros((frames, 118), np.float32)
    x0[:, 85] = 1.0
    return cams, gt, conf, x0


def demo_inputs():
    """BASELINE configs[0]: the reference's shipped demo - real cameras, keypoints, VPoser checkpoint and the
    reference's own initial guess, from the committed golden files (oracle/make_golden_demo.py)."""
    gd = os.path.join(ROOT, 'tests', 'golden')
    g = dict(np.load(os.path.join(gd, 'demo_fit_smpl.npz')))
    vpw = {k: v for k, v in np.load(os.path.join(gd, 'vposer_poser_epoch091_decoder.npz')).items() if k != 'source'}
    lsp = np.load(os.path.join(gd, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, kp_regressor=(lsp['rows'], lsp['cols'], lsp['vals']))
    cams = tuple(g[k].astype(np.float32) for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    x0 = np.zeros((1, 118), np.float32)
    x0[0, 85] = 1.0
    f = g['x0']                                  # reference order with VPoser: betas go transl scale embedding
    x0[0, 0:10] = f[0:10]; x0[0, 10:13] = f[10:13]; x0[0, 82:85] = f[13:16]; x0[0, 85] = f[16]; x0[0, 86:118] = f[17:49]
    return g, vpw, model, cams, g['gt_xy'][None].astype(np.float32), g['conf'][None].astype(np.float32), x0


def cpu_baseline(model, cams, gt, conf, stages, use_vposer, vpw, budget_s=15.0):
    """The PyTorch-CPU port of the reference closure + L-BFGS (oracle/closure_torch.py; the reference itself cannot
    travel to the GPU box): full 4-stage fits of the first frames of the same batch on ONE thread (the reference is
    fastest single-threaded) for ~budget_s; plus the raw closure (forward + backward) rate at 1 thread
    and at os.cpu_count() threads on a bounded number of calls (the all-threads run is an honest data point, not a
    recommendation: intra-op threading of these tiny tensor ops is slower than one thread)."""
    from oracle import closure_np as cn
    from oracle import closure_torch as ct
    lay, D = cn.param_layout(use_vposer)
    x0 = np.zeros(D)
    x0[lay['scale'][0]] = 1.0
    torch.set_num_threads(1)
    t0 = time.time()
    ncl = 0
    nfr = 0
    for b in range(gt.shape[0]):
        tc = ct.TorchClosure(model, cams, gt[b], conf[b], vposer=vpw)
        _, _, n = ct.fit_one(tc, x0, stages, use_vposer)
        ncl += n
        nfr += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    out = dict(value=round(ncl / dt, 1), unit='closures/s', cores=1, kind='port',
               sample='%d of the %d frames, full 4-stage fits (%d closures, %.1f s), PyTorch %s CPU port of the '
                      'reference closure + L-BFGS, 1 thread of %d host cores, %.0f ms/frame'
                      % (nfr, gt.shape[0], ncl, dt, torch.__version__, os.cpu_count(), 1e3 * dt / nfr))
    tc = ct.TorchClosure(model, cams, gt[0], conf[0], vposer=vpw)
    raw = {}
    for threads, calls in ((1, 60), (os.cpu_count() or 1, 6)):
        torch.set_num_threads(threads)
        tc.evaluate(x0, stages[0], use_vposer)                       # warm
        t1 = time.time()
        done = 0
        for _ in range(calls):
            tc.evaluate(x0, stages[0], use_vposer)
            done += 1
            if time.time() - t1 > 8.0:
                break
        raw[threads] = dict(closures_per_s=round(done / (time.time() - t1), 2), calls=done)
    torch.set_num_threads(1)
    out['raw_closure_fwd_bwd'] = {'threads_%d' % t: v for t, v in raw.items()}
    return out


def vertex_pass_variants(model, views, skin_topk):
    """Roofline of the vertex pass in its other instantiations, each on its own engine: the exact-fp32 contraction
    at 32 problems and the chunk-loop kernel at 128 problems (64 back-to-back launches inside one hipEvent pair)."""
    out = {}
    cams = syn.make_camera_ring(views)

    def measure(B, env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            eng = MvFit(model)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        rng = np.random.default_rng(B)
        x = np.zeros((B, 118), np.float32)
        x[:, :86] = rng.normal(0, 0.2, (B, 86))
        x[:, 85] = 1.0
        eng.set_problems(cams, np.zeros((B, views, 17, 2), np.float32), np.ones((B, views, 17), np.float32))
        eng.vertices(x)
        torch.cuda.synchronize()
        ms = min(eng.profile_vertex_pass_ms(64) for _ in range(3))
        eng.close()
        return ms
    for name, B, env, kern, key in (
            ('exact_fp32_B32', 32, {'MVFIT_EXACT_FP32': '1'}, 'lbs_vertex_pass_kernel<true>', None),
            ('split_fp16_B128', 128, {}, 'lbs_vertex_pass_split_loop_kernel<true>', 'B128'),
            ('half_basis_B32', 32, {'MVFIT_HALF_BASIS': '1'}, 'lbs_vertex_pass_split_kernel<true>', None),
            ('half_basis_B128', 128, {'MVFIT_HALF_BASIS': '1'}, 'lbs_vertex_pass_split_loop_kernel<true>', None)):
        ms = measure(B, env)
        nbytes = bytes_fwd(B, skin_topk) - (8557380 + 413400 if 'HALF' in ''.join(env) else 0)     # half-width posedirs + shapedirs
        ach = nbytes / (ms * 1e-3) / 1e9
        out[name] = dict(kernel=kern if skin_topk else kern.replace('<true>', '<false>'), problems=B,
                         avg_launch_us=round(ms * 1e3, 2), algorithmic_bytes=nbytes,
                         achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4),
                         traffic=pmc_traffic(key, 'split_loop') if key else None)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='configs1', choices=['configs1', 'demo'],
                    help="configs1: BASELINE configs[1] (default); demo: configs[0], the reference's shipped demo inputs")
    ap.add_argument('--frames', type=int, default=32, help='frames (problems) per GPU')
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--prior', default='l2', choices=['l2', 'vposer', 'gmm'])
    ap.add_argument('--sparse', action='store_true',
                    help='objective-vertices-only closure (no full vertex pass per closure)')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: --frames is the TOTAL, split over the ranks (default: weak, --frames per GPU)')
    ap.add_argument('--skin-topk', type=int, default=4,
                    help='non-zero skinning weights per vertex of the synthetic body (SMPL: <= 4); 0 = dense rows')
    ap.add_argument('--sdf', action='store_true',
                    help='configs[2]: SDF interpenetration term on (as wired: first triangle, grid 128; yaml coll_loss_weights)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-variants', action='store_true', help='skip the extra vertex-pass roofline variants')
    ap.add_argument('--dist-backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo for a dry run)')
    ap.add_argument('--single-device', action='store_true',
                    help='dry run of the multi-rank path on ONE GPU: every rank uses cuda:0 (needs --dist-backend gloo)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    demo = args.config == 'demo'
    if demo:
        assert world == 1, 'the demo is one frame'
        g, vpw, model, cams, gt, conf, x0 = demo_inputs()
        args.prior, args.views, args.skin_topk = 'vposer', 6, 0
        gmm = None
        B = total = 1
        lo = 0
    else:
        model = syn.make_body_model(0, skin_topk=args.skin_topk or None)
        vpw = syn.make_vposer_decoder() if args.prior == 'vposer' else None
        gmm = syn.make_gmm() if args.prior == 'gmm' else None
        total = args.frames if args.strong else args.frames * world
        lo, hi = shard_range(total, world, rank)       # contiguous global frames of this rank; seeds follow the global index
        B = hi - lo
        assert B > 0, 'more ranks than frames'
    eng = MvFit(model, vposer=vpw, gmm=None if gmm is None else syn.gmm_constants(gmm), device=local_rank)
    flags = 0
    if args.prior == 'vposer':
        flags |= _lib.F_VPOSER
    if args.prior == 'gmm':
        flags |= _lib.F_PRIOR_GMM
    if args.sparse:
        flags |= _lib.F_SPARSE_VERTS
    stages = stage_weights(1536.0, flags=flags, coll_w=[0.0, 0.0, 1000.0, 4500.0] if args.sdf else None)
    if args.sdf:
        eng.set_sdf(model['faces'], num_faces=1, grid_size=128)      # fit_smpl.yaml:55-59, fitting.py:367-368
    if demo:
        eng.set_problems(cams, gt, conf)
    else:
        cams, gt, conf, x0 = build_inputs(eng, B, args.views, seed0=1000 + lo)
    x0_d = torch.tensor(x0, device=dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1) if args.warmup else 0):
        xw, st = eng.fit(x0_d, stages)
        if world > 1:
            gather_results(xw, total)             # also sets up the RCCL rings outside the timed region
    # torch loads its own reduction / copy kernels lazily on first use (~80 ms): touch the exact ops of
    # the timed loop once here (also with --warmup 0), so that module loading is not billed to a fit
    _z = torch.zeros(B, device=dev, dtype=torch.int32)
    int(_z.sum().item()); int(_z.max().item())
    barrier()
    t0 = time.perf_counter()
    n_closure = 0
    n_iter = 0
    finals = None
    gathered = None
    n_max = 0
    counts = []
    passes = None
    busy = 0.0
    for _ in range(args.steps):
        tb = time.perf_counter()
        xf, st = eng.fit(x0_d, stages)             # returns when this rank's GPU has finished the fit
        busy += time.perf_counter() - tb
        if world > 1:
            gathered = gather_results(xf, total)   # the path's only collective (RCCL over xGMI), inside the step
        counts.append((st['n_closure'], st['n_iter']))      # device tensors: read after the timed region
        finals = st['final_loss']
        passes = st.get('passes')
    barrier()
    dt = time.perf_counter() - t0
    for ncl_t, nit_t in counts:
        n_closure += int(ncl_t.sum().item())
        n_iter += int(nit_t.sum().item())
    n_max = int(counts[-1][0].max().item())

    # max-over-ranks time, totals, per-rank busy time
    tot_closure, tot_iter, tmax = n_closure, n_iter, dt
    busy_all = [round(1e3 * busy / args.steps, 3)]
    if world > 1:
        import torch.distributed as dist
        red = torch.tensor([float(n_closure), float(n_iter)], device=dev, dtype=torch.float64)
        dist.all_reduce(red)
        tm = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        bz = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(bz, torch.tensor([busy / args.steps * 1e3], device=dev, dtype=torch.float64))
        busy_all = [round(float(b.item()), 3) for b in bz]
        assert gathered.shape[0] == total
        tot_closure, tot_iter, tmax = int(red[0].item()), int(red[1].item()), float(tm.item())

    # roofline of the dominant HBM kernel (LBS vertex pass), measured live with HIP events on the ctx stream: 64
    # back-to-back launches of the kernel exactly as the timed fits launched it, inside one event pair (a pair around a
    # single launch contains the markers' own 2-4 us)
    roof = None
    if not args.sparse and not demo:
        in_fit = bool(passes and passes['run'] > 0)          # asynchronous fit: ring operands, non-temporal streams
        nchunks = (B + 31) // 32
        kname = ('lbs_vertex_pass_split_kernel' if nchunks == 1 else 'lbs_vertex_pass_split_loop_kernel') + \
                ('<true>' if args.skin_topk and args.skin_topk <= 4 else '<false>')
        # (a) the kernel as it ran inside a fit: one more identical fit (outside the timed region) in which every pass
        #     launch carries an event pair stamped by the runtime with the dispatch's own begin / end (hipExtLaunchKernelGGL)
        eng.profile(True)
        eng.fit(x0_d, stages)
        pr = eng.profile_read()
        eng.profile(False)
        # (b) the kernel alone: 64 back-to-back launches inside one hipEvent pair, in the fit's launch flavour and with
        #     plain loads
        ms_b2b = min(eng.profile_vertex_pass_ms(64, as_in_async_fit=in_fit) for _ in range(3))
        ms_plain = min(eng.profile_vertex_pass_ms(64) for _ in range(3))
        ms_fit = pr['vertex_pass_ms'] if pr['vertex_pass_launches'] > 0 else ms_b2b
        ach = bytes_fwd(B, args.skin_topk) / (ms_fit * 1e-3) / 1e9
        roof = dict(bound='hbm', kernel=kname, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(ach / HBM_PEAK_GBS, 4),
                    traffic=pmc_traffic('B32' if B == 32 else ('B128' if B == 128 else ''), kname.split('<')[0])
                    if (args.views == 8 and args.skin_topk == 4) else None,
                    traffic_source='profiles/r2_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; tools/collect_profiles.sh)',
                    algorithmic_bytes=bytes_fwd(B, args.skin_topk), avg_launch_us=round(ms_fit * 1e3, 2),
                    mfma_util=mfma_util('B32' if B == 32 else ('B128' if B == 128 else ''), kname.split('<')[0], ms_fit * 1e3)
                    if (args.views == 8 and args.skin_topk == 4) else None,
                    mfma_util_note='SQ_VALU_MFMA_BUSY_CYCLES per launch (profiles/r2_pmc.json) / (avg_launch_us x 2.4 GHz x 1024 SIMDs)',
                    timed_region='every vertex-pass launch of one complete fit (%d launches), begin / end of each dispatch stamped by '
                                 'the runtime (hipExtLaunchKernelGGL events on the stream the kernel is launched on)' % pr['vertex_pass_launches'],
                    launch_flavour='asynchronous fit: operands from the ring, non-temporal basis stream and vertex stores, the optimiser '
                                   'kernel running concurrently on 32 other CUs' if in_fit else 'chained mode (plain loads, side outputs)',
                    alone_back_to_back_us=round(ms_b2b * 1e3, 2), alone_back_to_back_plain_loads_us=round(ms_plain * 1e3, 2),
                    frac_alone_plain_loads=round(bytes_fwd(B, args.skin_topk) / (ms_plain * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    launches_per_fit=(passes or {}).get('run'))
        if rank == 0 and not args.no_variants and world == 1:
            roof['variants'] = vertex_pass_variants(model, args.views, args.skin_topk)

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.prior != 'gmm':
            cpu_stages = [dict(s) for s in stages]
            cpu = cpu_baseline(model, cams, gt, conf, cpu_stages, args.prior == 'vposer', vpw)
            cpu['reference_in_survey_container'] = dict(
                value=list(SURVEY_REFERENCE_CLOSURES_PER_S), unit='closures/s',
                note='the imported reference itself (create_fitting_closure + LBFGSLs + run_fitting, PyTorch CPU, 1 and 8 '
                     'threads of 8 vCPUs) as timed in SURVEY.md section 6; it does not exist on the GPU box')
        fl = finals.cpu().numpy()
        if demo:
            workload = ('configs[0]: the reference demo (cfg_files/fit_smpl.yaml): 1 frame x 6 real views x 1 person, real keypoints / '
                        'cameras, VPoser decoder of the shipped checkpoint, synthetic body, 4 yaml stages')
        else:
            workload = ('%s: 1 person x %d views x %d synthetic frames%s, GMoF + pose prior (%s) + shape + angle priors, %s, 4 yaml stages'
                        % ('configs[2]' if args.sdf else 'configs[1]', args.views, total if args.strong else B,
                           ' in total' if args.strong else ' per GPU', args.prior,
                           'SDF term as wired (first triangle, grid 128, yaml coll_loss_weights)' if args.sdf else 'no SDF'))
        if args.sparse:
            mode = 'objective-vertices-only'
        elif passes and passes['run'] > 0:
            mode = ('full 6890-vertex pass per closure, asynchronous: one optimiser kernel publishes the operands of every trial '
                    'point, the pass of closure round r runs concurrently on the other CUs')
        else:
            mode = 'full 6890-vertex pass per closure, chained (pass -> step kernel per round)'
        out = {
            'metric': 'L-BFGS closure evaluations per second (fwd+bwd, all concurrently fitted problems), '
                      '%d-view 1-person 4-stage fits' % args.views,
            'value': round(tot_closure / tmax, 1), 'unit': 'closures/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * tmax / args.steps, 3),
            'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'dtype_note': 'all arithmetic fp32 (line-search scalars fp64 like the reference); the blendshape contraction of the '
                          'vertex pass takes its fp32 products as error-compensated split-fp16 pairs on the matrix pipe with '
                          'fp32 accumulation (vertices 5e-7 from the float64 oracle, as with the exact fp32 chain)',
            'data': ('real demo keypoints / cameras / VPoser checkpoint of the reference, seeded SMPL-shaped body' if demo else
                     'synthetic (seeded SMPL-shaped body%s, camera ring, noisy projected keypoints)'
                     % ((', %d skinning weights per vertex like SMPL' % args.skin_topk) if args.skin_topk else ', dense skinning rows')),
            'config': {'workload': workload, 'frames_per_gpu': B, 'frames_total': total, 'views': args.views, 'prior': args.prior,
                       'closure_mode': mode,
                       'parallelism': 'frame-sharded x%d, RCCL all_gather of results' % world},
            'ms_to_convergence_per_frame': round(1e3 * tmax / args.steps / max(B, 1), 4),
            'lbfgs_iters_per_s': round(tot_iter / tmax, 1),
            'closures_per_fit_per_frame': round(tot_closure / args.steps / total, 1),
            'closure_rounds_per_fit': n_max,      # = closures of the slowest frame of rank 0's batch
            'final_loss_median': float(np.median(fl)),
            'vertex_passes_last_fit': passes,
            'per_rank_busy_ms_per_step': busy_all,
            'roofline': roof, 'cpu_baseline': cpu,
        }
        if demo:
            out['reference_fit'] = dict(final_loss_fp32=float(g['fit_final32']), final_loss_fp64=float(g['fit_final64']),
                                        closures_fp32=int(g['fit_ncl32'].sum()), closures_fp64=int(g['fit_ncl64'].sum()),
                                        final_loss_spread_fp32=[float(v) for v in g['fit_spread32']],
                                        note='the reference itself on these inputs in the build container (oracle/make_golden_demo.py)')
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
