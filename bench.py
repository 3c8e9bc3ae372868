#!/usr/bin/env python
"""bench.py - closures/s of the device-resident multi-view SMPL fit on MI355X.

Metric (BASELINE.json): L-BFGS closure evaluations per second (forward + backward, summed over all
concurrently fitted problems) for 8-view, 1-person problems; ms to convergence per frame; the LBS
vertex pass as a fraction of the HBM roofline.

A "step" = ONE complete 4-stage fit (reference cfg_files/fit_smpl.yaml weights, L-BFGS lr=1,
max_iter=30, history=100, strong-Wolfe; outer maxiters=30, ftol=gtol=1e-9) of this rank's batch of
32 synthetic frames x 8 views x 1 person, every closure evaluating all 6890 vertices like the
reference does (return_verts=True).  Inputs are resident in HBM before the timed region.  Ranks
fit disjoint frames (weak scaling: 32 frames per GPU); the only collective is the final
all_gather of the fitted parameters over RCCL.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
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
# HBM-side bytes per vertex-pass launch at 32 problems from rocprofv3 PMC passes (profiles/r1c_kernel_stats_pmc.md, profiles/r1_pmc.md):
# 2 x FETCH_SIZE (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE.  Not measurable from inside
# this process; quoted only when the workload matches the profiled one.
PMC_TRAFFIC_B32 = 22.98e6          # dense skinning rows
PMC_TRAFFIC_B32_TOP4 = 22.53e6     # 4-sparse skinning rows (profiles/r1e_kernel_stats_pmc.md)


def bytes_fwd(B, skin_topk=0):
    """Algorithmic bytes of one LBS vertex pass over B problems; with
    k-sparse skinning weights the weight matrix is k (weight, joint) pairs per vertex instead of 24 floats."""
    const = CONST_BYTES if not skin_topk else CONST_BYTES - 661440 + 6890 * skin_topk * 8
    return const + PER_PROBLEM_BYTES * B


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


def cpu_baseline(model, cams, gt, conf, stages, use_vposer, vpw, budget_s=20.0):
    """The PyTorch-CPU port of the reference closure + L-BFGS (oracle/closure_torch.py), one thread
    (the reference is fastest single-threaded), on the first frames of the same batch."""
    from oracle import closure_np as cn
    from oracle import closure_torch as ct
    torch.set_num_threads(1)
    lay, D = cn.param_layout(use_vposer)
    t0 = time.time()
    ncl = 0
    nfr = 0
    for b in range(gt.shape[0]):
        tc = ct.TorchClosure(model, cams, gt[b], conf[b], vposer=vpw)
        x0 = np.zeros(D)
        x0[lay['scale'][0]] = 1.0
        _, _, n = ct.fit_one(tc, x0, stages, use_vposer)
        ncl += n
        nfr += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return dict(value=ncl / dt, unit='closures/s', cores=1, kind='port',
                sample='%d of the %d frames, full 4-stage fits (%d closures, %.1f s), PyTorch %s CPU '
                       'port of the reference closure + L-BFGS, 1 thread of %d host cores, %.0f ms/frame'
                       % (nfr, gt.shape[0], ncl, dt, torch.__version__, os.cpu_count(), 1e3 * dt / nfr))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=32, help='frames (problems) per GPU')
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--prior', default='l2', choices=['l2', 'vposer', 'gmm'])
    ap.add_argument('--sparse', action='store_true',
                    help='objective-vertices-only closure (no full vertex pass inside the loop)')
    ap.add_argument('--skin-topk', type=int, default=4,
                    help='non-zero skinning weights per vertex of the synthetic body (SMPL: <= 4); 0 = dense rows')
    ap.add_argument('--sdf', action='store_true',
                    help='configs[2]: SDF interpenetration term on (as wired: first triangle, grid 128; yaml coll_loss_weights)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
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

    model = syn.make_body_model(0, skin_topk=args.skin_topk or None)
    vpw = syn.make_vposer_decoder() if args.prior == 'vposer' else None
    gmm = syn.make_gmm() if args.prior == 'gmm' else None
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
    B = args.frames                                   # weak scaling: frames per GPU
    # rank r owns the contiguous global frames shard_range(B * world, world, r); seeds follow the
    # global frame index, so the job is the same set of frames however it is sharded
    lo, hi = shard_range(B * world, world, rank)
    assert hi - lo == B
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
            gather_results(xw, B * world)             # also sets up the RCCL rings outside the timed region
    # torch loads its own reduction / copy kernels lazily on first use (~80 ms): touch the exact ops of
    # the timed loop once here (also with --warmup 0), so that module loading is not billed to a fit
    _z = torch.zeros(B, device=dev, dtype=torch.int32)
    int(_z.sum().item()); int(_z.max().item())
    barrier()
    t0 = time.perf_counter()
    n_closure = 0
    n_iter = 0
    finals = None
    xf = None
    gathered = None
    n_max = 0
    passes = None
    for _ in range(args.steps):
        xf, st = eng.fit(x0_d, stages)
        if world > 1:
            gathered = gather_results(xf, B * world)  # the path's only collective (RCCL over xGMI), inside the step
        n_closure += int(st['n_closure'].sum().item())      # tiny D2H per step, after the fit finished
        n_iter += int(st['n_iter'].sum().item())
        n_max = int(st['n_closure'].max().item())
        finals = st['final_loss']
        passes = st.get('passes')
    barrier()
    dt = time.perf_counter() - t0

    # final gather over RCCL (the path's only collective) + max-over-ranks time
    tot_closure, tot_iter, tmax = n_closure, n_iter, dt
    if world > 1:
        import torch.distributed as dist
        red = torch.tensor([float(n_closure), float(n_iter)], device=dev, dtype=torch.float64)
        dist.all_reduce(red)
        tm = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        assert gathered.shape[0] == B * world
        tot_closure, tot_iter, tmax = int(red[0].item()), int(red[1].item()), float(tm.item())

    # roofline of the dominant kernel (LBS vertex pass): one more identical fit with per-launch
    # hipEvents on the ctx stream (kept out of the timed region: the event records perturb it)
    roof = None
    if not args.sparse:
        eng.profile(True)
        eng.fit(x0_d, stages)
        pr = eng.profile_read()
        eng.profile(False)
        # a hipEvent pair around ONE launch contains the markers' own few microseconds; the per-launch duration
        # quoted for the roofline is a region of 64 back-to-back launches of the same kernel on the pose operands
        # the fit left, inside one event pair (agrees with rocprofv3 --kernel-trace, profiles/); the in-pipeline
        # single-launch bracket is reported next to it
        raw_vp_ms = pr['vertex_pass_ms']
        pr['vertex_pass_ms'] = eng.profile_vertex_pass_ms(64)
        if pr['vertex_pass_launches'] > 0:
            ach = bytes_fwd(B, args.skin_topk) / (pr['vertex_pass_ms'] * 1e-3) / 1e9
            roof = dict(bound='hbm', kernel='lbs_vertex_pass_kernel', achieved=round(ach, 1),
                        peak=HBM_PEAK_GBS, unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4),
                        traffic=(PMC_TRAFFIC_B32 if args.skin_topk == 0 else PMC_TRAFFIC_B32_TOP4) if (B == 32 and args.views == 8 and args.skin_topk in (0, 4)) else None,
                        traffic_source='profiles/r1e_kernel_stats_pmc.md / r1c_kernel_stats_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)',
                        algorithmic_bytes=bytes_fwd(B, args.skin_topk), avg_launch_us=round(pr['vertex_pass_ms'] * 1e3, 2),
                        avg_launch_us_single_bracketed=round(raw_vp_ms * 1e3, 2), timed_region='64 back-to-back launches, one hipEvent pair',
                        launches=pr['vertex_pass_launches'],
                        step_kernel_avg_us=round(pr['step_ms'] * 1e3, 2))

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu_stages = [dict(s) for s in stages]
            cpu = cpu_baseline(model, cams, gt, conf, cpu_stages, args.prior == 'vposer', vpw) \
                if args.prior != 'gmm' else None
        fl = finals.cpu().numpy()
        out = {
            'metric': 'L-BFGS closure evaluations per second (fwd+bwd, all concurrently fitted problems), '
                      '8-view 1-person 4-stage fits',
            'value': round(tot_closure / tmax, 1), 'unit': 'closures/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * tmax / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'dtype_note': 'all arithmetic fp32 (line-search scalars fp64 like the reference); the blendshape contraction of the '
                          'vertex pass takes its fp32 products as error-compensated split-fp16 pairs on the matrix pipe with '
                          'fp32 accumulation (vertices 5e-7 from the float64 oracle, as with the exact fp32 chain)',
            'data': 'synthetic (seeded SMPL-shaped body%s, camera ring, noisy projected keypoints)' % ((', %d skinning weights per vertex like SMPL' % args.skin_topk) if args.skin_topk else ', dense skinning rows'),
            'config': {'workload': '%s: 1 person x %d views x %d synthetic frames per GPU, '
                                   'GMoF + pose prior (%s) + shape + angle priors, %s, 4 yaml stages'
                                   % ('configs[2]' if args.sdf else 'configs[1]', args.views, B, args.prior,
                                      'SDF term as wired (first triangle, grid 128, yaml coll_loss_weights)' if args.sdf
                                      else 'no SDF'),
                       'frames_per_gpu': B, 'views': args.views, 'prior': args.prior,
                       'closure_mode': 'objective-vertices-only' if args.sparse else 'full 6890-vertex pass per closure',
                       'parallelism': 'frame-sharded x%d, RCCL all_gather of results' % world},
            'ms_to_convergence_per_frame': round(1e3 * tmax / args.steps / B, 4),
            'lbfgs_iters_per_s': round(tot_iter / tmax, 1),
            'closures_per_fit_per_frame': round(tot_closure / args.steps / (B * world), 1),
            'closure_rounds_per_fit': n_max,      # = closures of the slowest frame of rank 0's batch (the batch advances in lock-step rounds)
            'final_loss_median': float(np.median(fl)),
            'vertex_passes_last_fit': passes,
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
