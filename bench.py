#!/usr/bin/env python
"""bench.py -- FLAIR volumes/sec of sliding-window inference (160x192x160, 128^3 patches, step 0.5,
Gaussian aggregation) on N MI355X.  One "step" = one whole volume through the hot path on every rank:
volume H2D -> extract patches -> Generic_UNet forward -> seg head + softmax + mirror average + Gaussian
accumulate -> normalise + argmax -> label-map D2H (SURVEY.md 8d's timing protocol).  Prints ONE JSON line.

  python bench.py                         # N=1, configs[1] (fp32 tensors / accumulate, products from fp16 pairs; mirroring on): the headline
  python bench.py --precision fp32        # the exact-fp32 kernels (v_mfma_f32_32x32x2_f32)
  python bench.py --gpus N                # spawns the N ranks itself (fresh child processes, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N    # or under torchrun
  python bench.py --precision bf16 --storage bf16      # configs[2]: bf16 forward, a step = 8 volumes

Multi-GPU: volumes are independent, so ranks shard the cohort (one volume per rank per step, no data-path
collective): weak scaling.  `--shard patches` is the patch-sharded mode of configs[3] (one volume per
step, RCCL all-gather of the weighted tiles).  `--tta 0` drops nnU-Net's 8-fold mirroring (the
reference's `--disable_tta` stage-II predictions); the default keeps it, as `DeepWMH_predict` does.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters (dense peaks).  f16x2: the stride-1 pair kernels form three of the four
# partial products of the fp16 hi/lo split, i.e. three 16-bit MFMAs' worth of k slots per tap and 16 channels: ceiling = 16-bit peak / 3
# (until r03x all four products were formed and the ceiling was peak / 4 = 625: `peak_basis` says which one a line uses).
PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0, 'f16x2': 2500.0 / 3.0}
PEAK_BASIS = {'fp32': 'v_mfma_f32_32x32x2_f32 dense peak', 'bf16': '16-bit MFMA dense peak', 'f16x2': '16-bit MFMA dense peak / 3 (three partial products)'}
NOMINAL_SCLK_MHZ = 2400            # the clock the guide's peaks are quoted at
HBM_PEAK_GBS = 8000.0
FWD_GFLOP_PER_PATCH = 954.46     # BASELINE.md section 2 (the engine's own count of what it queued is reported next to it)
AGG_GBYTES_PER_VOLUME = 0.46     # aggregation read-modify-write + volume in / label map out
KERNEL_NAME = {
    'fp32': 'conv3d_k3_ws_kernel<cfg0>',
    'bf16': 'conv3d_k3_bf16_ws_kernel<cfg0>',
    'f16x2': 'conv3d_k3_bf16_ws_kernel<cfg0, PAIR>',
    'bf16_storage': 'cl_conv3_kernel (every stride-1 launch)',
}


def patches_per_forward(storage):
    """Patches in one forward of the sliding-window predictor.  With mirroring the 8 mirrored copies of a tile are a batch of 8; 16 puts two
    tiles' copies into one forward, which gives the deep levels (16^3 and below) grids that fill the chip: +1.8 % in the default mode (32: +1 %,
    64: -0.2 %, the activation tensors then outgrow the Infinity Cache), +2.3 % in the channels-last mode (same-box A/B: 12.55 -> 12.84 volumes/s)."""
    return int(os.environ.get('WMH_BENCH_BATCH', '16'))


def mode_key(precision, storage):
    return precision if precision in ('fp32', 'f16x2') else ('bf16_storage' if storage == 'bf16' else 'bf16')


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--tta', type=int, default=1, help='1: 8-fold mirroring as DeepWMH_predict; 0: --disable_tta')
    ap.add_argument('--shape', type=int, nargs=3, default=[160, 192, 160])
    ap.add_argument('--cpu-baseline', choices=['auto', 'full', 'sample', 'off'], default='auto',
                    help='auto: time the oracle on the WHOLE volume without mirroring when a probe says that fits in '
                         '~60 s (it does on the GPU box), else on one patch; full / sample force one of the two')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='same as --cpu-baseline off')
    ap.add_argument('--precision', choices=['fp32', 'bf16', 'f16x2'], default='f16x2',
                    help='f16x2 (default, the headline): fp32 tensors and accumulation, every product from fp16 pairs (21-22 significant '
                         'bits per operand; softmax within 1e-5 of the fp32 CPU oracle at full size); fp32 = the exact-fp32 kernels; '
                         'bf16 = BASELINE.json configs[2] (bf16 matrix-core operands, fp32 accumulate)')
    ap.add_argument('--storage', choices=['fp32', 'bf16'], default='fp32',
                    help='bf16 (with --precision bf16): activation tensors are kept as bf16; not the headline')
    ap.add_argument('--volumes-per-step', type=int, default=0,
                    help='volumes each rank predicts per step (default 1; 8 with --precision bf16 = configs[2])')
    ap.add_argument('--shard', choices=['volumes', 'patches'], default='volumes',
                    help='volumes: one volume per rank per step (weak scaling, no collective); patches: ONE volume per '
                         'step, its patches split over the ranks with an RCCL all-gather of the weighted tiles (strong)')
    ap.add_argument('--also', choices=['auto', 'off'], default='auto',
                    help='auto (N=1, fp32 headline only): after the headline, measure configs[2] (channels-last bf16, 8 volumes per '
                         'step), the exact-fp32 kernels and one training iteration (fp32, f16x2) and report them in an "also" object')
    ap.add_argument('--cli', choices=['auto', 'off'], default='auto',
                    help='auto (with --also auto): after everything else, run bin/nnUNet_predict (with mirroring; with --disable_tta --save_softmax) '
                         'and bin/nnUNet_train as child processes on BASELINE-sized cases and report their rates in `cli` / `summary.cli`')
    ap.add_argument('--legs', choices=['auto', 'off', 'only'], default='auto',
                    help='the two collective paths north_star names, on the ranks of this run, after the headline: configs[3] (ONE volume, its patches over '
                         'the ranks, RCCL all_gather_into_tensor of the weighted tiles) and configs[4] (f16x2 training iterations, gradient all-reduce under '
                         'the backward pass) -> `summary.configs3_patch_sharded` / `summary.configs4_ddp`.  auto: at N > 1 in-process; at N = 1 in a child '
                         'process whose ONE-rank RCCL group forces every collective (the rehearsal).  only: nothing but the two legs')
    ap.add_argument('--detail-file', default='auto',
                    help='where the FULL record goes (every mode with its clock samples, phases, parity objects, sample descriptions); the one '
                         'printed line carries the numbers only.  auto: gpurun_out/bench_detail.json when gpurun_out/ exists, else nowhere')
    ap.add_argument('--dry-run', action='store_true',
                    help='launch plumbing only (CPU test of --gpus N): gloo process group, no kernels, value null')
    args = ap.parse_args(argv)
    if args.no_cpu_baseline:
        args.cpu_baseline = 'off'
    if args.volumes_per_step <= 0:
        args.volumes_per_step = 8 if args.precision == 'bf16' else 1
    return args


# ------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` from a bare shell
# ------------------------------------------------------------------------------------------------------------
def self_launch(argv):
    """Parent of a multi-GPU run.  Nothing in this process touches the GPU (no torch import either): it starts
    `torch.distributed.run` as a CHILD, which starts one fresh rank per GPU, relays their output and returns
    the child's exit code.  (Replacing a GPU-initialised process with exec is what takes boxes down.)"""
    n = parse_args(argv).gpus
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle as the checker and as the reported baseline; never in the timed GPU region)
# ------------------------------------------------------------------------------------------------------------
def cpu_baseline(plan, sd, vol, tta, mode, gpu_notta, gpu_patch_logits, extra_notta=None):
    """Times the CPU oracle (kind 'port': our torch-CPU restatement of nnU-Net; the reference's own nnU-Net
    fork is not available).  'full': the whole volume without mirroring -- 8 patch forwards + Gaussian
    aggregation + argmax, after one warm-up forward -- which is the complete path at 1/8 of the mirrored
    forward count; the mirrored figure is that time with the forward part x8.  'sample': one patch forward.
    gpu_notta: (seg, prob) numpy of the HIP path on the same volume without mirroring, or None.
    extra_notta: {mode name: (seg, prob)} of other fp32-grade modes to check against the same oracle run (returned as a
    third value, {mode name: parity dict})."""
    import torch
    from oracle import sliding_window as osw
    from oracle.generic_unet import unet_from_plan
    cores = torch.get_num_threads()                 # torch's default for this host (physical cores it may use)
    net = unet_from_plan(plan).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net.do_ds = False
    P = plan.patch_size
    x = torch.from_numpy(np.ascontiguousarray(vol[None, :, :P[0], :P[1], :P[2]]))
    sub = x[:, :, :P[0] // 2, :P[1] // 2, :P[2] // 2]
    npatch, mirrors, parity, parity_extra = 8, (8 if tta else 1), None, {}
    with torch.no_grad():
        net(sub)                                    # cold: oneDNN primitive creation
        t0 = time.time()
        net(sub)
        t_probe = time.time() - t0                  # 1/8 of a patch forward, warm (scales badly on many-core hosts: a bound only)

    def sm(a):
        e = np.exp(a - a.max(1, keepdims=True))
        return e / e.sum(1, keepdims=True)

    ref_patch = t_patch = None
    if mode != 'sample' or 8 * t_probe <= 120.0:
        with torch.no_grad():
            t0 = time.time()
            ref_patch = net(x).numpy()              # one whole patch forward: the sample, and the warm-up of the full run
            t_patch = time.time() - t0
    if mode == 'auto':                              # whole volume when it fits in about two minutes of host time
        mode = 'full' if npatch * t_patch <= 120.0 else 'sample'

    if mode == 'full':
        fwd_s = [0.0]
        orig = osw.mirror_and_predict

        def timed_mirror(*a, **k):
            t = time.time()
            r = orig(*a, **k)
            fwd_s[0] += time.time() - t
            return r
        osw.mirror_and_predict = timed_mirror
        try:
            t0 = time.time()
            seg_o, prob_o = osw.predict_3d_tiled(net, vol, P, 0.5, False, True)
            t_full = time.time() - t0
        finally:
            osw.mirror_and_predict = orig
        t_est = t_full + (mirrors - 1) * fwd_s[0]
        out = dict(value=1.0 / t_est, unit='volumes/s', cores=cores, kind='port',
                   sample='whole volume WITHOUT mirroring timed end to end (%.1f s: %d forwards %.1f s + aggregation %.1f s); value = 1 / (that with the '
                          'forwards x %d for mirroring)' % (t_full, npatch, fwd_s[0], t_full - fwd_s[0], mirrors),
                   timed_no_mirroring=dict(value=1.0 / t_full, seconds=round(t_full, 2), forward_seconds=round(fwd_s[0], 2)))
        margin = np.abs(prob_o[1] - prob_o[0])

        def versus_oracle(seg_g, prob_g):
            return dict(sample='whole volume, no mirroring, vs the CPU oracle',
                        max_abs_softmax_err=float(np.abs(prob_g - prob_o).max()),
                        argmax_mismatches=int(((seg_g != seg_o) & (margin > 2e-4)).sum()),
                        argmax_mismatches_any_margin=int((seg_g != seg_o).sum()),
                        dice_vs_oracle_mask=float(osw.hard_dice_binary(seg_o, seg_g)))
        if gpu_notta is not None:
            parity = versus_oracle(*gpu_notta)
        parity_extra = {k: versus_oracle(*v) for k, v in (extra_notta or {}).items()}
    else:
        if t_patch is not None:
            t, frac = t_patch, 1.0
            what = 'one %dx%dx%d patch forward (1/%d of the volume)' % (P[0], P[1], P[2], npatch * mirrors)
        else:
            t, frac = t_probe, 0.125
            what = 'one %dx%dx%d sub-patch forward (1/8 patch), scaled x8' % (P[0] // 2, P[1] // 2, P[2] // 2)
        est = npatch * mirrors * t / frac
        out = dict(value=1.0 / est, unit='volumes/s', cores=cores, kind='port',
                   sample='%s, %.1f s, forward only, extrapolated to %d forwards' % (what, t, npatch * mirrors))
    if parity is None and ref_patch is not None and gpu_patch_logits is not None:
        margin = np.abs(ref_patch[:, 1] - ref_patch[:, 0])
        parity = dict(sample='first patch, no mirroring',
                      max_abs_softmax_err=float(np.abs(sm(gpu_patch_logits) - sm(ref_patch)).max()),
                      argmax_mismatches=int(((gpu_patch_logits.argmax(1) != ref_patch.argmax(1)) & (margin > 4e-4)).sum()))
    return out, parity, parity_extra


# ------------------------------------------------------------------------------------------------------------
# shader clock and socket power over a timed region (the 16-bit MFMA modes run below the nominal clock: DESIGN.md 3.2)
# ------------------------------------------------------------------------------------------------------------
class ClockSampler:
    """Reads the device's hwmon `freq1_input` (shader clock, Hz) and `power1_input` (socket power, uW) every 50 ms from a thread
    while a timed region runs.  Plain sysfs reads: no GPU call, no child process.  Reports medians; None where the files are absent."""

    def __init__(self, device_index=0):
        import glob
        self.freq = self.power = None
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = '%04x:%02x:%02x' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cands = []
        for f in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input')):
            dev = os.path.realpath(os.path.join(os.path.dirname(f), '..', '..'))
            cands.append((os.path.basename(dev), f))
        pick = [f for (d, f) in cands if want and d.startswith(want)]
        if not pick and len(cands) == 1:
            pick = [cands[0][1]]
        if pick:
            self.freq = pick[0]
            pw = os.path.join(os.path.dirname(pick[0]), 'power1_input')
            self.power = pw if os.path.exists(pw) else None
        self.f, self.w, self._stop, self._th = [], [], False, None

    def _run(self):
        while not self._stop:
            try:
                self.f.append(int(open(self.freq).read()) / 1e6)
                if self.power:
                    self.w.append(int(open(self.power).read()) / 1e6)
            except Exception:
                pass
            time.sleep(0.05)

    def __enter__(self):
        if self.freq:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join(timeout=1.0)

    def report(self):
        if not self.f:
            return None
        med = lambda v: sorted(v)[len(v) // 2]
        return dict(sclk_mhz_median=round(med(self.f)), sclk_mhz_min=round(min(self.f)), sclk_mhz_nominal=NOMINAL_SCLK_MHZ,
                    socket_power_w_median=round(med(self.w)) if self.w else None, samples=len(self.f))


def add_sustained(roofline, clock):
    """roofline.frac stays priced against the guide's peak (2.4 GHz); this adds what the fraction is at the clock the chip held."""
    if clock and roofline and roofline.get('bound') == 'mfma' and clock['sclk_mhz_median'] > 0:
        k = min(1.0, clock['sclk_mhz_median'] / NOMINAL_SCLK_MHZ)
        roofline['frac_at_sustained_clock'] = round(roofline['frac'] / k, 4)


def in_kernel_clock_f16x2(device, batch):
    """One stamped launch of the dominant kernel on the dominant layer (32 -> 32 channels at 128^3, `batch` patches), outside the timed
    region: the kernel's own s_memtime stamps (shader-clock cycles of a consumer wave's MFMA phases, barrier waits and epilogues, which
    add up to the workgroup's life time) over the launch's duration give the clock the CUs actually ran at -- the hwmon figure in `clock`
    is the power management's target, below which the clock is stretched while the matrix pipes are busy -- and the share of those
    cycles in which an MFMA could issue at all."""
    import ctypes as C
    import torch
    from deepwmh_amd import _lib
    lib, P = _lib.load_diag(), _lib.ptr          # the product library's kernels hold no stamp code (csrc/wmh_common.h)
    N, S, Cin, Cout = int(batch), 128, 32, 32
    cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(Cin, Cout, S, S, S, 1), Cout, S)
    if not lib.wmh_conv3d_k3_f16x2_supported(cfg):
        return None
    g = torch.Generator(device=device).manual_seed(1)
    x = torch.randn(N, Cin, S, S, S, device=device, generator=g)
    ss = torch.stack([1 + 0.3 * torch.randn(N, Cin, device=device, generator=g), 0.2 * torch.randn(N, Cin, device=device, generator=g)], -1).contiguous()
    w = (torch.randn(Cout, Cin, 27, device=device, generator=g) / (27 * Cin) ** 0.5).contiguous()
    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), dtype=torch.int16, device=device)
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2_device(cfg, P(w), Cin, Cout, 0, Cin, 0, P(wf), None), 'pack')
    y = torch.empty(N, Cout, S, S, S, device=device)
    part = torch.empty(N, Cout, lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, S, S, S), 4, device=device)

    def launch():
        _lib.check(lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x), P(ss), Cin, None, None, 0, P(wf), None, P(y), P(part), None, N, S, S, S, Cout, 0.01, None), 'f16x2')
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=device)
    lib.wmh_debug_set_conv_stamps(C.c_void_p(stamps.data_ptr()))
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        torch.cuda.synchronize()
    finally:
        lib.wmh_debug_set_conv_stamps(None)
    ms = e0.elapsed_time(e1)
    st = stamps.view(-1, 8).cpu().numpy().astype('float64')
    st = st[st[:, 3] > 0]
    if ms <= 0 or len(st) == 0:
        return None
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    cycles = float((st[:, 0] + st[:, 1] + st[:, 2]).mean())
    mhz = cycles / ms / 1e3
    flop = 2.0 * 27 * Cin * Cout * N * S ** 3
    issue = (82.0 / 27.0) * flop / 32768.0 / (4 * ncu) * 32.0   # 82 MFMAs per 27 taps and 16 channels; 32 cycles per v_mfma_f32_32x32x16_f16, per SIMD
    tflops = flop / ms / 1e9
    return dict(layer='32 -> 32 channels, %d x 128^3 (one stamped launch after the timed region)' % N, launch_ms=round(ms, 3), achieved=round(tflops, 1),
                sclk_mhz=round(mhz), cycles_per_workgroup=round(cycles), mfma_issue_cycles=round(issue), mfma_issue_share=round(issue / cycles, 3),
                phases=dict(mfma=round(float(st[:, 0].mean() / cycles), 3), barrier_wait=round(float(st[:, 1].mean() / cycles), 3),
                            epilogue=round(float(st[:, 2].mean() / cycles), 3)),
                frac_at_in_kernel_clock=round(tflops / (PEAK_TFLOPS['f16x2'] * mhz / NOMINAL_SCLK_MHZ), 3))


# ------------------------------------------------------------------------------------------------------------
# roofline block of the dominant kernel, per precision mode
# ------------------------------------------------------------------------------------------------------------
def roofline_block(timings, precision, storage, dt):
    """timings: (cfg, start_event, end_event, algorithmic_flop, algorithmic_bytes) per launch of the dominant
    kernel, HIP events recorded on the launch stream inside the timed region."""
    tot_ms = sum(e0.elapsed_time(e1) for (_, e0, e1, _, _) in timings)
    tot_flop = sum(f for (_, _, _, f, _) in timings)
    tot_bytes = sum(b for (_, _, _, _, b) in timings)
    nl = max(1, len(timings))
    secs = tot_ms * 1e-3
    tflops = tot_flop / secs / 1e12 if secs > 0 else 0.0
    gbs = tot_bytes / secs / 1e9 if secs > 0 else 0.0
    peak_t = PEAK_TFLOPS[precision]
    # the roof that bounds this kernel = the larger of its two floor times
    bound = 'mfma' if tot_flop / (peak_t * 1e12) >= tot_bytes / (HBM_PEAK_GBS * 1e9) else 'hbm'
    traffic, traffic_note = None, None
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            t = json.load(f)
        t = t.get(mode_key(precision, storage), t if precision == 'fp32' else {})
        traffic = t.get('traffic_bytes_per_launch')     # PMC counters cannot be read live: separate rocprofv3 passes
        traffic_note = t.get('measured')
    except Exception:
        pass
    r = dict(bound=bound, kernel=KERNEL_NAME[mode_key(precision, storage)])
    if bound == 'mfma':
        r.update(achieved=round(tflops, 3), peak=round(peak_t, 1), peak_basis=PEAK_BASIS[precision], unit='TFLOP/s', frac=round(tflops / peak_t, 4))
    else:
        r.update(achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4))
    r.update(traffic=traffic, traffic_measured=traffic_note, launches=len(timings), avg_launch_ms=round(tot_ms / nl, 4),
             algorithmic_gflop_per_launch=round(tot_flop / nl / 1e9, 3),
             algorithmic_gbytes_per_launch=round(tot_bytes / nl / 1e9, 4),
             mfma_frac=round(tflops / peak_t, 4), hbm_frac=round(gbs / HBM_PEAK_GBS, 4),
             share_of_step_time=round(secs / dt, 4) if dt > 0 else None)
    assert r['frac'] <= 1.0, 'roofline fraction above 1: wrong peak for this mode'
    return r


def workload_string(args, tta, forwards, world):
    shp = '%dx%dx%d' % tuple(args.shape)
    mir = 'mirroring %s' % ('on' if tta else 'off')
    if args.precision == 'fp32':
        return 'configs[1], exact-fp32 kernels: one %s volume per GPU, 128^3 patches, step 0.5, Gaussian aggregation, %s, %d forwards per volume' % (shp, mir, forwards)
    if args.precision == 'f16x2':
        return ('configs[1]: one %s volume per GPU, 128^3 patches, step 0.5, fp32 tensors and accumulation (3x3x3 products from fp16 hi+lo pairs), '
                'Gaussian aggregation, %s, %d forwards per volume' % (shp, mir, forwards))
    return ('configs[2]: bf16 forward (%s tensors), fused InstanceNorm+LeakyReLU, batch of %d %s volumes per step per GPU, %s, %d forwards per volume'
            % ('bf16' if args.storage == 'bf16' else 'fp32', args.volumes_per_step, shp, mir, forwards))


# ------------------------------------------------------------------------------------------------------------
# the other modes, on the same record (N = 1): exact fp32, configs[2], one training iteration
# ------------------------------------------------------------------------------------------------------------
TRAIN_TFLOP_PER_ITER = 5.727       # batch 2 x 128^3, forward + backward = 3 x 2 x 954.55 GFLOP


def measure_inference_mode(plan, sd, device, precision, storage, V, steps, warmup, tta, shape, ref32=None):
    """One more inference mode with the headline's timing protocol (H2D + hot path + D2H per volume, HIP events on the
    dominant kernel's launches).  ref32: (seg, prob) device tensors of the fp32 HIP path on volume 0, for the accuracy figures."""
    import torch
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from deepwmh_amd.synthetic import synthetic_flair
    eng = UNetEngine(plan, sd, device, precision=precision, storage=storage)
    pred = SlidingWindowPredictor(eng, do_mirroring=tta, batch=patches_per_forward(storage))
    vols_h = [torch.from_numpy(synthetic_flair(tuple(shape), seed=v)).pin_memory() for v in range(V)]
    seg_h = torch.empty(tuple(shape), dtype=torch.uint8).pin_memory()
    vol_d = torch.empty((1,) + tuple(shape), dtype=torch.float32, device=device)

    def run_step():
        for v in range(V):
            vol_d.copy_(vols_h[v], non_blocking=True)
            out = pred.predict_device(vol_d, return_prob=True)
            seg_h.copy_(out[0], non_blocking=True)
        return out

    for _ in range(warmup):
        run_step()
    eng.timed_cfgs, eng.timings, eng.acct = {0}, [], dict(flop=0.0, bytes=0.0)
    torch.cuda.synchronize()
    with ClockSampler(torch.cuda.current_device()) as cs:
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    eng.timed_cfgs = set()
    acct, eng.acct = eng.acct, None
    forwards = 8 * (8 if tta else 1)
    nvol = steps * V
    gb_vol = acct['bytes'] / nvol / 1e9 + AGG_GBYTES_PER_VOLUME
    out = dict(volumes_per_step=V, steps=steps, warmup=warmup, ms_per_step=round(1e3 * dt / steps, 3), value=round(nvol / dt, 4),
               unit='volumes/s', end_to_end_tflops=round(forwards * FWD_GFLOP_PER_PATCH / 1e3 * nvol / dt, 2),
               hbm_whole_path=dict(algorithmic_gbytes_per_volume=round(gb_vol, 2), achieved_gbs=round(gb_vol * nvol / dt, 1),
                                   peak_gbs=HBM_PEAK_GBS, frac=round(gb_vol * nvol / dt / HBM_PEAK_GBS, 4)),
               roofline=roofline_block(eng.timings, precision, storage, dt), clock=cs.report())
    add_sustained(out['roofline'], out['clock'])
    if ref32 is not None:
        vol_d.copy_(vols_h[0])
        s16, p16 = pred.predict_device(vol_d, return_prob=True)
        s32, p32 = ref32
        inter = float(((s16 > 0) & (s32 > 0)).sum())
        out['parity_vs_fp32_path'] = dict(
            max_abs_softmax_diff=float((p16 - p32).abs().max()), voxels_changing_class=int((s16 != s32).sum()), voxels=int(s32.numel()),
            dice_vs_fp32_mask=2 * inter / (float((s16 > 0).sum()) + float((s32 > 0).sum()) + 1e-6))
    return out, eng, vol_d, vols_h


def measure_no_mirroring_rate(plan, sd, device, shape, steps=6):
    """volumes/s of the hot path without mirroring (8 forwards per volume in one batch of 8; `--disable_tta`), same protocol as the headline:
    the kernel-only figure the `--disable_tta --save_softmax` command is put next to."""
    import torch
    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from deepwmh_amd.synthetic import synthetic_flair
    eng = UNetEngine(plan, sd, device, precision='f16x2')
    pred = SlidingWindowPredictor(eng, do_mirroring=False, batch=16)
    vol_h = torch.from_numpy(synthetic_flair(tuple(shape), seed=0)).pin_memory()
    seg_h = torch.empty(tuple(shape), dtype=torch.uint8).pin_memory()
    vol_d = torch.empty((1,) + tuple(shape), dtype=torch.float32, device=device)

    def step():
        vol_d.copy_(vol_h, non_blocking=True)
        seg_h.copy_(pred.predict_device(vol_d, return_prob=True)[0], non_blocking=True)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    del eng, pred
    torch.cuda.empty_cache()
    return round(steps / dt, 3)


def measure_training_mode(plan, sd, device, precision, iters, warmup):
    """One nnUNetTrainerV2 iteration (forward with deep supervision, Dice+CE, backward, clip, SGD-Nesterov) at batch 2 x 128^3,
    BASELINE plan, synthetic patches and labels resident in HBM."""
    import torch
    from deepwmh_amd.trainer import UNetTrainer, downsample_targets
    tr = UNetTrainer(plan, sd, device, precision=precision)
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn(2, 1, 128, 128, 128, generator=g).to(device)
    lab = (torch.rand(2, 128, 128, 128, generator=g) < 0.05).to(torch.uint8).to(device)
    td = downsample_targets(lab, plan.num_pool, plan.pool_op_kernel_sizes)
    losses = []
    for _ in range(warmup):
        losses.append(tr.run_iteration(x, td, 1e-2).clone())
    torch.cuda.synchronize()
    with ClockSampler(torch.cuda.current_device()) as cs:
        t0 = time.perf_counter()
        for _ in range(iters):
            losses.append(tr.run_iteration(x, td, 1e-2).clone())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    tflops = TRAIN_TFLOP_PER_ITER * iters / dt
    peak = PEAK_TFLOPS[precision]
    losses = [float(v) for v in torch.stack(losses).cpu()]
    del tr
    out = dict(workload='one nnUNetTrainerV2 iteration (forward with deep supervision, Dice+CE, backward, clip, SGD-Nesterov), batch 2 x 128^3, BASELINE plan',
               precision=precision, steps=iters, warmup=warmup, ms_per_step=round(1e3 * dt / iters, 3), value=round(2 * iters / dt, 3), unit='patches/s',
               loss_first_to_last=[round(losses[0], 6), round(losses[-1], 6)], clock=cs.report(),
               # whole iteration against the mode's forward ceiling: an upper bound for f16x2 / bf16, whose strided and transposed kernels form
               # four products or stay fp32
               roofline=dict(bound='mfma', kernel='whole iteration', achieved=round(tflops, 2), peak=round(peak, 1), peak_basis=PEAK_BASIS[precision],
                             unit='TFLOP/s', frac=round(tflops / peak, 4), algorithmic_tflop_per_step=TRAIN_TFLOP_PER_ITER))
    add_sustained(out['roofline'], out['clock'])
    return out


def training_parity_tiny():
    """The training step against FLOAT64 torch autograd (the oracle's restatement of nnUNetTrainerV2.run_iteration) on the committed tiny
    network: loss and every parameter gradient of one iteration, per mode.  The checker, not the thing measured.  Median and worst
    parameter error in max-norm and the cosine of the whole gradient: the fixture has a voxel whose pre-activation is ~1e-6 from a
    LeakyReLU kink, so an evaluation that rounds it to the other side is 5 % off on that channel's parameters without being wrong
    (tests/test_gpu_trainer.py::test_f16x2_training_step_matches_autograd_oracle); the median is the robust figure."""
    import copy
    import torch
    from deepwmh_amd.plan import tiny_plan
    from deepwmh_amd.trainer import UNetTrainer, downsample_targets
    from oracle import training as ot
    from oracle.generic_unet import unet_from_plan
    plan = tiny_plan(patch=32, base=4, num_pool=3, max_feat=32)
    sd = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'tiny_net_weights.npz')))
    io = np.load(os.path.join(ROOT, 'tests', 'golden', 'tiny_net_io.npz'))
    vol = io['volume'][0]
    data = np.stack([vol[4:36, 8:40, 4:36], vol[8:40, 0:32, 2:34][::-1].copy()])[:, None].astype(np.float32)
    labels = np.stack([(io['seg_tta'][4:36, 8:40, 4:36] > 0), np.random.default_rng(5).random((32, 32, 32)) < 0.2]).astype(np.uint8)
    net = unet_from_plan(plan)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = copy.deepcopy(net).double().train()
    net.do_ds = True
    weights = ot.deep_supervision_weights(plan.num_pool)
    x = torch.from_numpy(data)
    tgt = [t.double() for t in ot.downsample_targets(torch.from_numpy(labels[:, None].astype(np.float32)), plan.num_pool)]
    loss = ot.multiple_output_loss(net(x.double()), tgt, weights)
    loss.backward()
    out = {}
    for precision in ('fp32', 'f16x2'):
        tr = UNetTrainer(plan, sd, 'cuda:0', precision=precision)
        tr.forward(x.cuda())
        l = float(tr.backward(downsample_targets(torch.from_numpy(labels).cuda(), plan.num_pool)).cpu())
        num = den = dot = 0.0
        rows = []
        for name, p in net.named_parameters():
            ref = p.grad if p.grad is not None else torch.zeros_like(p)
            got = tr.g[name].double().cpu()
            num += float(((got - ref) ** 2).sum())
            den += float((ref ** 2).sum())
            dot += float((got * ref).sum())
            if float(ref.abs().max()) > 1e-5:
                rows.append(float((got - ref).abs().max()) / float(ref.abs().max()))
        rows.sort()
        gn = float(torch.cat([tr.g[n].double().cpu().reshape(-1) for n, _ in net.named_parameters()]).norm())
        out[precision] = dict(sample='tiny net, batch 2 x 32^3, one iteration vs float64 autograd', loss=l, oracle_loss=float(loss.detach()), gradient_rel_l2_err=(num / max(den, 1e-30)) ** 0.5,
                              gradient_cosine=dot / max(gn * den ** 0.5, 1e-30),
                              median_parameter_rel_max_err=rows[len(rows) // 2], worst_parameter_rel_max_err=rows[-1])
    return out

# ------------------------------------------------------------------------------------------------------------
# the boundary DeepWMH sees: the three commands as fresh child processes
# ------------------------------------------------------------------------------------------------------------
def measure_cli(plan, sd, shape, n_mirror=12, n_softmax=40, train_batches=40):
    """`bin/nnUNet_predict` and `bin/nnUNet_train` exactly as /root/reference/deepwmh/main/predict.py:153-156 and
    deepwmh/pipeline/DCNN_multistage.py:292-309,331-344 run them -- child processes found on PATH, talking through folders of .nii.gz
    files -- on BASELINE-sized synthetic cases: prediction with mirroring (DeepWMH_predict), prediction with `--disable_tta --save_softmax`
    (the 5 x N Stage-II predictions), one epoch of `-b <train_batches>` iterations.  Children of this process (which has released its device
    memory but keeps its context: they share the GPU with an idle parent); never an exec.  Rates are the commands' own report lines:
    `cases/s` of the whole case loop (reads, preprocessing, GPU, export, compressed writes; model load excluded) and the rate once the first
    case is through; the command's wall-clock is next to them."""
    import re
    import shutil
    import tempfile
    import torch
    from deepwmh_amd import checkpoint as ck, nifti
    from deepwmh_amd.synthetic import synthetic_flair
    tmp = tempfile.mkdtemp(prefix='wmh_bench_cli_')
    out = {}
    try:
        env = dict(os.environ, PATH=os.path.join(ROOT, 'bin') + os.pathsep + os.environ.get('PATH', ''), CUDA_VISIBLE_DEVICES=os.environ.get('CUDA_VISIBLE_DEVICES', '0'))
        env.pop('MI355WMH_GPUS', None)
        # -- a model folder in the layout DeepWMH installs, and input cases
        res_dir, task = os.path.join(tmp, 'model'), 'Task002_FinalModel'
        folder = ck.trainer_folder(res_dir, task)
        ck.save_checkpoint(folder, {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()},
                           ck.make_plans(plan, spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True), 'all', 'model_best', epoch=1)
        img = os.path.join(tmp, 'in')
        os.makedirs(img)
        ncase = max(n_mirror, n_softmax)
        base = [synthetic_flair(tuple(shape), seed=100 + c)[0] for c in range(4)]
        for c in range(ncase):                                  # (four distinct volumes, cycled: the files are what is timed, not their content)
            v = base[c % 4]
            nifti.write(os.path.join(img, 'case%02d_0000.nii.gz' % c), (v * 50 + 300 * (v != 0)).astype(np.float32), spacing=(1.0, 1.0, 1.0))
        common = ('-i %s -o %%s -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 -t %s -f all -chk model_best --disable_post_processing '
                  '--selected_cases %%s' % (img, task))

        def run(cmd, extra_env=None):
            t0 = time.time()
            try:
                p = subprocess.run(cmd.split(), env=dict(env, **(extra_env or {})), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            except subprocess.TimeoutExpired as e:
                return 124, 'timed out after 600 s: %s' % (e.stdout or '')[-300:], time.time() - t0
            return p.returncode, p.stdout, time.time() - t0

        def predict_leg(n, flags, outdir):
            cases = ' '.join('case%02d' % c for c in range(n))
            rc, log, wall = run('nnUNet_predict ' + common % (os.path.join(tmp, outdir), cases) + flags, dict(RESULTS_FOLDER=res_dir))
            m = re.search(r'(\d+) case\(s\) in ([0-9.]+) s \(([0-9.]+) cases/s(?:; first case done after ([0-9.]+) s, then ([0-9.]+) cases/s)?', log)
            if rc != 0 or not m:
                return dict(error='rc %d: %s' % (rc, log[-300:]))
            gpu = [float(v) for v in re.findall(r'gpu ([0-9.]+) s', log)]
            nfiles = len(os.listdir(os.path.join(tmp, outdir)))
            return dict(cases=int(m.group(1)), cases_per_s=float(m.group(3)), first_case_s=float(m.group(4) or 0), cases_per_s_after_first=float(m.group(5) or 0),
                        gpu_s_per_case_median=sorted(gpu)[len(gpu) // 2] if gpu else None, command_wall_s=round(wall, 2), files_written=nfiles)
        out['predict_mirroring'] = predict_leg(n_mirror, '', 'out_tta')
        out['predict_disable_tta_save_softmax'] = predict_leg(n_softmax, ' --disable_tta --save_softmax', 'out_softmax')
        # -- training: raw dataset -> nnUNet_plan_and_preprocess -> one epoch of nnUNet_train
        raw, pre, res2 = os.path.join(tmp, 'raw'), os.path.join(tmp, 'pre'), os.path.join(tmp, 'results')
        root = os.path.join(raw, 'nnUNet_raw_data', 'Task001_LabelDenoising')
        os.makedirs(os.path.join(root, 'imagesTr')); os.makedirs(os.path.join(root, 'labelsTr'))
        training = []
        for i in range(4):
            v, name = base[i], 'case_%02d' % i
            nifti.write(os.path.join(root, 'imagesTr', name + '_0000.nii.gz'), (v * 40 + 200 * (v != 0)).astype(np.float32), spacing=(1.0, 1.0, 1.0))
            nifti.write(os.path.join(root, 'labelsTr', name + '.nii.gz'), (v > 1.6).astype(np.float32), spacing=(1.0, 1.0, 1.0))
            training.append({'image': os.path.join(root, 'imagesTr', name + '.nii.gz'), 'label': os.path.join(root, 'labelsTr', name + '.nii.gz')})
        with open(os.path.join(root, 'dataset.json'), 'w') as f:
            json.dump({'name': '', 'description': '', 'reference': '', 'license': '', 'release': 'v1.0', 'tensorImageSize': '3D', 'modality': {'0': 'Modality_01'},
                       'labels': {'0': 'background', '1': 'lesion'}, 'numTraining': 4, 'numTest': 0, 'training': training, 'test': []}, f)
        tenv = dict(nnUNet_raw_data_base=raw, nnUNet_preprocessed=pre, RESULTS_FOLDER=res2)
        rc, log, wall_pp = run('nnUNet_plan_and_preprocess -t 1 --verify_dataset_integrity', tenv)
        if rc != 0:
            out['train'] = dict(error='plan_and_preprocess rc %d: %s' % (rc, log[-300:]))
        else:
            rc, log, wall = run('nnUNet_train 3d_fullres nnUNetTrainerV2 1 all -e 2 -b %d --noval --save_every_epoch --disable_postprocessing_on_folds --val_batches 1' % train_batches, tenv)
            ep = re.findall(r'epoch (\d+)/\d+: .*\(([0-9.]+) s: (\d+) iterations at ([0-9.]+) ms, validation ([0-9.]+) s\)', log)
            if rc != 0 or len(ep) < 2:
                out['train'] = dict(error='rc %d: %s' % (rc, log[-300:]))
            else:
                # epoch 2 = the steady state (epoch 1 carries buffer allocation and kernel loading); its wall-clock holds the iterations, the
                # validation batch and whatever of the three checkpoint files' writing the background writer did not hide
                e2 = ep[1]
                out['train'] = dict(batches_per_epoch=int(e2[2]), ms_per_iteration_in_epoch=float(e2[3]), epoch_wall_s=float(e2[1]),
                                    ms_per_iteration_all_in=round(1e3 * float(e2[1]) / int(e2[2]), 2), first_epoch_wall_s=float(ep[0][1]),
                                    command_wall_s=round(wall, 2), plan_and_preprocess_wall_s=round(wall_pp, 2))
    except Exception as e:                                      # the boundary leg must not take the headline down
        out['error'] = '%s: %s' % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


# ------------------------------------------------------------------------------------------------------------
# the two collective paths of north_star on the ranks of this run (BASELINE.json configs[3], configs[4])
# ------------------------------------------------------------------------------------------------------------
class _DryTileBackend:
    """CPU stand-in of distributed.HipTileBackend for `--dry-run`: the geometry is the real one (steps, Gaussian, patch order), a tile is a
    cheap deterministic function of the volume window -- enough to prove the exchange, the fixed accumulation order and the ragged last round."""

    def __init__(self, patch):
        self.patch = tuple(patch)

    def geometry(self, V):
        import torch
        from deepwmh_amd.sliding_window import compute_steps_for_sliding_window, get_gaussian
        A = tuple(max(v, p) for v, p in zip(V, self.patch))
        self.pad_lo = tuple((a - v) // 2 for a, v in zip(A, V))
        st = compute_steps_for_sliding_window(self.patch, A, 0.5)
        origins = [(x, y, z) for x in st[0] for y in st[1] for z in st[2]]
        self.gauss = torch.from_numpy(get_gaussian(self.patch)) if len(origins) > 1 else None
        self.A = A
        return A, self.pad_lo, origins, 2, self.patch

    def zeros(self, shape):
        import torch
        return torch.zeros(shape, dtype=torch.float32)

    def tile(self, vol, origin):
        import torch
        P, (x, y, z) = self.patch, origin
        pad = torch.zeros((vol.shape[0],) + self.A)
        lo = self.pad_lo
        pad[:, lo[0]:lo[0] + vol.shape[1], lo[1]:lo[1] + vol.shape[2], lo[2]:lo[2] + vol.shape[3]] = vol
        w = pad[0, x:x + P[0], y:y + P[1], z:z + P[2]]
        p1 = torch.sigmoid(3.0 * w + 0.01 * (x + 2 * y + 3 * z))
        t = torch.stack([1.0 - p1, p1])
        return t * self.gauss if self.gauss is not None else t

    def accumulate(self, agg, tile, origin):
        P, (x, y, z) = self.patch, origin
        agg[:, x:x + P[0], y:y + P[1], z:z + P[2]] += tile

    def count_map(self, A, origins):
        import torch
        cnt, P = torch.zeros(A), self.patch
        for (x, y, z) in origins:
            cnt[x:x + P[0], y:y + P[1], z:z + P[2]] += self.gauss if self.gauss is not None else 1.0
        return cnt

    def finalize(self, agg, cnt, pad_lo, V):
        sl = tuple(slice(l, l + v) for l, v in zip(pad_lo, V))
        prob = (agg / cnt)[(slice(None),) + sl]
        import torch
        return prob.argmax(0).to(torch.uint8), prob


def _predicted(section, world):
    """profiles/r05g_shard_cost.json: what ONE rank does was measured on one MI355X (round 5), the wire was modelled at 100 / 153 GB/s per xGMI link."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r05g_shard_cost.json')) as f:
            return json.load(f)[section]['predicted'].get('W%d' % world)
    except Exception:
        return None


def multi_gpu_legs(args, rank, world, device, dry=False, volumes=10, iters=8):
    """configs[3] and configs[4] with the headline's protocol (barrier + device synchronisation on both sides, MAX over ranks).  At world size 1
    the collectives are forced through the one-rank group (`force_collective`): the RCCL kernels run, nothing crosses a link.  Every rank
    calls this; the dict is complete on rank 0."""
    import torch
    import torch.distributed as dist
    from deepwmh_amd.distributed import OverlappedGradientAverager, PatchShardedPredictor
    force = world == 1
    cuda = not dry

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    def timed(fn, n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if cuda else 'cpu')
        if dist.is_initialized() and world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    out = {}
    # ---- configs[3]: ONE volume at a time, patch r, r + W, ... on rank r, one all_gather_into_tensor per round --------------------------
    if dry:
        shape, patch = (40, 48, 40), (32, 32, 32)
        backend = _DryTileBackend(patch)
        vol = torch.randn((1,) + shape, generator=torch.Generator().manual_seed(0))
        serial = PatchShardedPredictor.__new__(PatchShardedPredictor)      # the serial loop: a one-rank predictor without a group
        serial.b, serial.group, serial.force, serial.rank, serial.world, serial.stall_marks = backend, None, False, 0, 1, None
        serial_predict = lambda: serial.predict(vol)
    else:
        from deepwmh_amd.distributed import HipTileBackend
        from deepwmh_amd.network import UNetEngine
        from deepwmh_amd.plan import baseline_plan
        from deepwmh_amd.sliding_window import SlidingWindowPredictor
        from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
        plan = baseline_plan()
        sd = he_init_state_dict(plan, seed=0)
        eng = UNetEngine(plan, sd, device, precision='f16x2')
        pred = SlidingWindowPredictor(eng, do_mirroring=True, batch=8)     # a rank's round = the 8 mirrored copies of ONE tile
        backend = HipTileBackend(pred)
        vol = torch.from_numpy(synthetic_flair(tuple(args.shape), seed=0)).to(device)
        serial_predict = lambda: pred.predict_device(vol, return_prob=True)
    sharded = PatchShardedPredictor(backend, force_collective=force)
    seg, prob = sharded.predict(vol)                                        # warm-up (exchange buffers, RCCL channels)
    seg, prob = seg.clone(), prob.clone()
    identical = None
    if rank == 0:
        s0, p0 = serial_predict()
        identical = bool(torch.equal(s0, seg) and torch.equal(p0, prob))
    sharded.stall_marks = []
    dt = timed(lambda: sharded.predict(vol), volumes)
    exposed = sharded.exposed_exchange_ms() / volumes
    sharded.stall_marks = None
    ntile = len(backend.geometry(tuple(vol.shape[1:]))[2])
    tile_bytes = 4 * 2 * int(np.prod(backend.patch if dry else pred.patch))
    out['configs3_patch_sharded'] = dict(
        world=dist.get_world_size() if dist.is_initialized() else 1, volumes=volumes, volumes_per_s=round(volumes / dt, 3), ms_per_volume=round(1e3 * dt / volumes, 2),
        exposed_exchange_ms_per_volume_rank0=round(exposed, 3), rounds_per_volume=-(-ntile // world), all_gather_bytes_per_round=world * tile_bytes,
        bit_identical_to_serial_rank0=identical, forced_one_rank_collective=force, predicted=_predicted('configs3_patch_sharded', world),
        scaling='strong', dry_run=dry or None)
    if not dry:
        del sharded, backend, pred, eng
        torch.cuda.empty_cache()
    # ---- configs[4]: data-parallel training iterations, gradient all-reduce in >= 8 MB runs under the backward pass ---------------------
    if dry:
        sizes = [40, 8, 400, 12, 1000, 4, 36]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        ranges = {'p%d' % i: (int(offs[i]), int(offs[i + 1])) for i in range(len(sizes))}
        g = torch.zeros(int(offs[-1]))
        av = OverlappedGradientAverager(g, ranges, None, min_elems=300, force_collective=force)
        base = torch.arange(int(offs[-1]), dtype=torch.float32)

        def iteration():
            g.copy_(base * (rank + 1))
            for names in (['p6', 'p5'], ['p4'], ['p1'], ['p2', 'p3'], ['p0']):
                av.ready(names)
            av.finish()
        iteration()
        ok = bool(torch.equal(g, base * (sum(range(1, world + 1)) / world)))
        dt = timed(iteration, iters)
        nbytes, nmsg, losses = 4 * g.numel(), 3, None
    else:
        from deepwmh_amd.trainer import UNetTrainer, downsample_targets
        tr = UNetTrainer(plan, sd, device, precision='f16x2')
        gen = torch.Generator(device='cpu').manual_seed(rank)               # every rank its own two patches
        x = torch.randn(2, 1, 128, 128, 128, generator=gen).to(device)
        lab = (torch.rand(2, 128, 128, 128, generator=gen) < 0.05).to(torch.uint8).to(device)
        td = downsample_targets(lab, plan.num_pool, plan.pool_op_kernel_sizes)
        nmsg = [0]

        def iteration():
            tr.forward(x)
            loss = tr.backward(td, None, force_collective=force)
            nmsg[0] = getattr(tr._avg, 'last_messages', 0)
            tr.optimizer_step(1e-2)
            return loss
        losses = [float(iteration().cpu()) for _ in range(2)]
        dt = timed(iteration, iters)
        # the parameters must be the same on every rank after the averaged steps
        chk = tr.params.double().sum().reshape(1)
        ok = True
        if world > 1:
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            ok = bool((lo == hi).item())
        nbytes, nmsg = 4 * int(tr.grads.numel()), nmsg[0]
        del tr
        torch.cuda.empty_cache()
    out['configs4_ddp'] = dict(
        world=dist.get_world_size() if dist.is_initialized() else 1, iterations=iters, ms_per_iteration=round(1e3 * dt / iters, 3), patches_per_s=round(2 * world * iters / dt, 2),
        all_reduce_bytes_per_iteration=nbytes, all_reduce_messages_per_iteration=nmsg, parameters_identical_on_all_ranks=ok,
        forced_one_rank_collective=force, predicted=_predicted('configs4_data_parallel_training', world), scaling='weak',
        loss_first_two=losses, dry_run=dry or None)
    return out


def legs_child():
    """The N = 1 form of the two legs: a fresh child process whose first GPU act is a ONE-rank RCCL group (the parent keeps its context; never an
    exec).  Returns the child's dict or {'error': ...}."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--legs', 'only', '--gpus', '1'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=float(os.environ.get('WMH_BENCH_LEGS_TIMEOUT_S', '420')))
    except subprocess.TimeoutExpired:                     # (the rehearsal must never take the headline down with it)
        return dict(error='the one-rank RCCL child did not finish in time')
    for ln in reversed(p.stdout.splitlines()):
        if ln.startswith('LEGS '):
            return json.loads(ln[5:])
    return dict(error='rc %d: %s' % (p.returncode, p.stdout[-300:]))



def legs_summary(legs):
    """The two legs' figures for the printed line: numbers only (the full dicts go to the detail file).  `pred`: the round-5 cost model's
    figure for this world size at 100 / 153 GB/s per link (profiles/r05g_shard_cost.json), None at world size 1."""
    if 'error' in legs:
        return dict(legs_error=legs['error'][-200:])
    c3, c4 = legs['configs3_patch_sharded'], legs['configs4_ddp']
    p3, p4 = c3.get('predicted') or {}, c4.get('predicted') or {}
    return dict(
        configs3_patch_sharded=dict(world=c3['world'], volumes_per_s=c3['volumes_per_s'], exposed_exchange_ms=c3['exposed_exchange_ms_per_volume_rank0'],
                                    bit_identical=c3['bit_identical_to_serial_rank0'], forced=c3['forced_one_rank_collective'],
                                    pred=[v['volumes_per_s'] for k, v in sorted(p3.items()) if k.startswith('link_')] or None),
        configs4_ddp=dict(world=c4['world'], ms_per_iteration=c4['ms_per_iteration'], patches_per_s=c4['patches_per_s'],
                          all_reduce_mbytes=round(c4['all_reduce_bytes_per_iteration'] / 1e6, 1), same_params=c4['parameters_identical_on_all_ranks'],
                          pred_ms=[v['iteration_ms'] for k, v in sorted(p4.items()) if k.startswith('link_')] or None))


def compact_line(res):
    """The ONE printed line.  The driver keeps the last 2000 characters of stdout, so: the contract's keys, numbers only (prose lives in
    DESIGN.md, the full record in --detail-file), and LAST a `summary` that holds every mode's headline figure."""
    def pick(d, *keys):
        return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}
    r = res.get('roofline') or {}
    also = res.get('also') or {}
    line = pick(res, 'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling')
    line['vs_baseline'] = res.get('vs_baseline')
    line.update(pick(res, 'dtype', 'data'))
    line['config'] = pick(res.get('config', {}), 'workload', 'tta', 'patch', 'step', 'batch', 'volumes_per_step', 'weights', 'parallelism')
    rl = pick(r, 'bound', 'kernel', 'achieved', 'peak', 'peak_basis', 'unit', 'frac')
    rl['traffic'] = r.get('traffic')
    if r.get('traffic') is not None:
        rl['traffic_source'] = 'separate rocprofv3 --pmc passes (profiles/pmc_traffic.json), not this run'
    rl.update(pick(r, 'launches', 'avg_launch_ms', 'algorithmic_gflop_per_launch', 'algorithmic_gbytes_per_launch', 'share_of_step_time'))
    ik = r.get('in_kernel')
    if isinstance(ik, dict) and 'sclk_mhz' in ik:
        rl['in_kernel'] = pick(ik, 'sclk_mhz', 'mfma_issue_share', 'phases')
    line['roofline'] = rl
    if res.get('cpu_baseline'):
        line['cpu_baseline'] = pick(res['cpu_baseline'], 'value', 'unit', 'cores', 'kind', 'sample')
    if res.get('clock'):
        line['clock'] = pick(res['clock'], 'sclk_mhz_median', 'socket_power_w_median')
    line.update(pick(res, 'end_to_end_tflops'))
    if res.get('hbm_whole_path'):
        line['hbm_whole_path_frac'] = res['hbm_whole_path'].get('frac')

    rnd = lambda v, n=4: None if v is None else float(('%.' + str(n) + 'g') % v)
    summary = {}
    name = {'f16x2': 'configs1_f16x2', 'fp32': 'configs1_fp32_exact'}.get(res.get('precision'), 'configs2_bf16')
    summary[name] = rnd(res.get('value'))
    frac, pmax, mism, sclk = {name: r.get('frac')}, {}, {}, {name: (res.get('clock') or {}).get('sclk_mhz_median')}
    if 'fp32_exact' in also:
        summary['configs1_fp32_exact'] = rnd(also['fp32_exact'].get('value'))
        frac['configs1_fp32_exact'] = also['fp32_exact']['roofline'].get('frac')
        sclk['configs1_fp32_exact'] = (also['fp32_exact'].get('clock') or {}).get('sclk_mhz_median')
        po = also['fp32_exact'].get('parity_vs_oracle')
        if po:
            pmax['fp32_exact_vs_oracle'] = rnd(po['max_abs_softmax_err'], 3)
            mism['fp32_exact_vs_oracle'] = [po['argmax_mismatches'], po['argmax_mismatches_any_margin']]
    if 'configs2_bf16_storage' in also:
        c2 = also['configs2_bf16_storage']
        summary['configs2_bf16'] = rnd(c2.get('value'))
        frac['configs2_bf16'] = c2['roofline'].get('frac')
        sclk['configs2_bf16'] = (c2.get('clock') or {}).get('sclk_mhz_median')
        pf = c2.get('parity_vs_fp32_path')
        if pf:
            pmax['configs2_bf16_vs_fp32_path'] = rnd(pf['max_abs_softmax_diff'], 3)
            summary['configs2_dice_vs_fp32_mask'] = rnd(pf['dice_vs_fp32_mask'], 6)
    tm, tpar = {}, {}
    for prec in ('fp32', 'f16x2'):
        t = also.get('train_' + prec)
        if t:
            tm[prec] = rnd(t['ms_per_step'])
            frac['train_' + prec] = t['roofline'].get('frac')
            if t.get('parity'):
                tpar[prec] = rnd(t['parity']['gradient_rel_l2_err'], 3)
    if tm:
        summary['train_ms'] = tm
    if tpar:
        summary['train_gradient_rel_l2_vs_float64_tiny_net'] = tpar
    par = res.get('parity')
    if par and 'max_abs_softmax_err' in par:
        pmax[('f16x2' if res.get('precision') == 'f16x2' else res.get('precision', 'mode')) + '_vs_oracle'] = rnd(par['max_abs_softmax_err'], 3)
        if 'argmax_mismatches_any_margin' in par:
            mism[('f16x2' if res.get('precision') == 'f16x2' else res.get('precision', 'mode')) + '_vs_oracle'] = [par['argmax_mismatches'], par['argmax_mismatches_any_margin']]
    pf = res.get('parity_vs_fp32_path')
    if pf:
        pmax[name + '_vs_fp32_path'] = rnd(pf['max_abs_softmax_diff'], 3)
    if pmax:
        summary['parity_max_abs'] = pmax
    if mism:
        summary['argmax_mismatches_outside_2e-4_margin_and_any'] = mism
    summary['frac'] = {k: v for k, v in frac.items() if v is not None}
    summary['sclk_mhz'] = {k: v for k, v in sclk.items() if v is not None}
    if res.get('cpu_baseline'):
        summary['cpu_volumes_per_s'] = rnd(res['cpu_baseline']['value'], 3)
    cli = res.get('cli')
    if cli:
        # the commands DeepWMH runs, as child processes: [cases/s of the whole case loop, cases/s after the first case, kernel-only volumes/s];
        # training: [ms per iteration inside an epoch, epoch wall-clock / iterations (validation batch + checkpoint files included), kernel-only ms]
        c, ko = {}, cli.get('kernel_only', {})
        pm, ps, tr = cli.get('predict_mirroring', {}), cli.get('predict_disable_tta_save_softmax', {}), cli.get('train', {})
        if 'cases_per_s' in pm:
            c['predict_mirroring'] = [rnd(pm['cases_per_s']), rnd(pm['cases_per_s_after_first']), rnd(ko.get('mirroring_volumes_per_s'))]
        if 'cases_per_s' in ps:
            c['predict_disable_tta_save_softmax'] = [rnd(ps['cases_per_s']), rnd(ps['cases_per_s_after_first']), rnd(ko.get('no_mirroring_volumes_per_s'))]
        if 'ms_per_iteration_all_in' in tr:
            c['train_ms'] = [rnd(tr['ms_per_iteration_in_epoch']), rnd(tr['ms_per_iteration_all_in']), rnd(ko.get('train_ms_f16x2'))]
        errs = [k for k, v in cli.items() if isinstance(v, dict) and 'error' in v] + (['cli'] if 'error' in cli else [])
        if errs:
            c['failed'] = errs
        summary['cli'] = c
    if res.get('legs'):
        summary.update(legs_summary(res['legs']))
    line['summary'] = summary                      # last key: it must survive the driver's 2000-character tail
    return line


# ------------------------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        print('bench.py: WORLD_SIZE=%d overrides --gpus %d' % (world, args.gpus), file=sys.stderr)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

    if args.dry_run:                                   # plumbing test: no device, no kernels
        if world > 1:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.01)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        legs = multi_gpu_legs(args, rank, world, 'cpu', dry=True, volumes=2, iters=2) if (world > 1 and args.legs != 'off') else None
        if rank == 0:
            res = dict(metric='FLAIR volumes/sec sliding-window inference (160x192x160)', value=None,
                       unit='volumes/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                       ms_per_step=1e3 * dt / args.steps, dry_run=True)
            if legs:
                res['summary'] = legs_summary(legs)
            print(json.dumps(res), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    from deepwmh_amd.network import UNetEngine
    from deepwmh_amd.plan import baseline_plan
    from deepwmh_amd.sliding_window import SlidingWindowPredictor
    from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair

    # WMH_BENCH_BACKEND=gloo: the ranks may share GPUs (rank r on device r mod the visible count) -- the rehearsal of `--gpus N` on a box with
    # fewer than N GPUs (numbers meaningless, plumbing and bit-identity real); the default, nccl = RCCL, wants one GPU per rank
    backend = os.environ.get('WMH_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)                   # before the process group: RCCL binds to the current device
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world)
    device = 'cuda:%d' % dev_index

    if args.legs == 'only':
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group(backend, rank=0, world_size=1)      # one-rank RCCL group: every collective is forced through it
        legs = multi_gpu_legs(args, rank, world, device)
        if rank == 0:
            print('LEGS ' + json.dumps(legs), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return 0

    plan = baseline_plan()
    sd = he_init_state_dict(plan, seed=0)
    eng = UNetEngine(plan, sd, device, precision=args.precision, storage=args.storage)
    tta = bool(args.tta)
    pred = SlidingWindowPredictor(eng, do_mirroring=tta, batch=patches_per_forward(args.storage))
    by_patch = args.shard == 'patches'
    V = 1 if by_patch else args.volumes_per_step
    vols_h = [torch.from_numpy(synthetic_flair(tuple(args.shape), seed=(0 if by_patch else rank * V + v))).pin_memory()
              for v in range(V)]
    seg_h = torch.empty(tuple(args.shape), dtype=torch.uint8).pin_memory()
    vol_d = torch.empty((1,) + tuple(args.shape), dtype=torch.float32, device=device)
    if by_patch:
        from deepwmh_amd.distributed import HipTileBackend, PatchShardedPredictor
        sharded = PatchShardedPredictor(HipTileBackend(pred))

    def run_step():
        """The boundary hands over host volumes and takes host label maps: both copies are part of the step."""
        out = None
        for v in range(V):
            vol_d.copy_(vols_h[v], non_blocking=True)                  # H2D, stream ordered
            out = sharded.predict(vol_d) if by_patch else pred.predict_device(vol_d, return_prob=True)
            seg_h.copy_(out[0], non_blocking=True)                     # D2H of the label map
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run_step()
    eng.timed_cfgs = {0}                               # dominant kernel: conv configuration 0 of either precision
    eng.timings = []
    # algorithmic work queued in the timed region, counted by the engine per launch with the tensor formats of this mode
    # (a layer reads its input once, writes its output once, + weights; fp32 mode: 4.17 GB per patch forward)
    eng.acct = dict(flop=0.0, bytes=0.0)
    barrier()
    with ClockSampler(torch.cuda.current_device()) as cs:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            seg, prob = run_step()
        barrier()
        dt = time.perf_counter() - t0
    clock = cs.report()
    eng.timed_cfgs = set()
    acct, eng.acct = eng.acct, None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline = roofline_block(eng.timings, args.precision, args.storage, dt)
    add_sustained(roofline, clock)
    if rank == 0 and args.precision == 'f16x2' and roofline is not None:
        try:
            ik = in_kernel_clock_f16x2(device, patches_per_forward(args.storage))
            if ik:
                roofline['in_kernel'] = ik
        except Exception as e:                          # diagnostics only
            roofline['in_kernel'] = dict(error=str(e))

    legs = None
    if world > 1 and args.legs == 'auto':
        # the collective paths on the same ranks, after the headline's timed region (its buffers are released first)
        eng._bufs.clear()
        torch.cuda.empty_cache()
        try:
            legs = multi_gpu_legs(args, rank, world, device)
        except Exception as e:                          # (every rank raises or none: the legs' collectives are symmetric)
            legs = dict(error='%s: %s' % (type(e).__name__, e))

    if rank == 0:
        forwards = 8 * (8 if tta else 1)
        nvol = args.steps if by_patch else world * args.steps * V
        vols_rank = args.steps * V                      # volumes this rank queued in the timed region
        gb_vol = acct['bytes'] / vols_rank / 1e9 + AGG_GBYTES_PER_VOLUME
        dtype = ('f32' if args.precision == 'fp32' else
                 'f32 (tensors, accumulation; 3x3x3 products from fp16 hi+lo pairs, 21-22 significant bits)' if args.precision == 'f16x2' else
                 'bf16 operands / f32 accumulate+storage' if args.storage == 'fp32' else
                 'bf16 operands and tensors / f32 accumulate')
        res = dict(metric='FLAIR volumes/sec sliding-window inference (160x192x160)', precision=args.precision, value=nvol / dt,
                   unit='volumes/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling='strong' if by_patch else 'weak',
                   vs_baseline=None, dtype=dtype, data='synthetic',
                   config=dict(workload=workload_string(args, tta, forwards, world),
                               tta=tta, patch=[128, 128, 128], step=0.5, batch=patches_per_forward(args.storage), volumes_per_step=V,
                               weights='random He-init (seed 0)',
                               timed_region='volume H2D (19.7 MB, pinned) + hot path + label-map D2H (4.9 MB) per volume',
                               parallelism='patches of one volume sharded over ranks, RCCL all-gather of weighted tiles' if by_patch
                               else 'volumes sharded over ranks, no collective'),
                   end_to_end_tflops=round(forwards * FWD_GFLOP_PER_PATCH / 1e3 * nvol / dt, 2),
                   counted_gflop_per_patch_forward=round(acct['flop'] / vols_rank / forwards / 1e9, 2),
                   # north_star's "fraction of the HBM roofline" for the whole path (algorithmic bytes of this mode)
                   hbm_whole_path=dict(algorithmic_gbytes_per_volume=round(gb_vol, 2),
                                       achieved_gbs=round(gb_vol * nvol / dt, 1), peak_gbs=HBM_PEAK_GBS,
                                       frac=round(gb_vol * nvol / dt / HBM_PEAK_GBS, 4)),
                   roofline=roofline, clock=clock)
        if args.precision == 'bf16' or (args.precision == 'f16x2' and args.also == 'off'):
            # accuracy of the mode against the exact-fp32 HIP path on the same volume (same weights)
            eng32 = UNetEngine(plan, sd, device)
            pred32 = SlidingWindowPredictor(eng32, do_mirroring=tta, batch=8)
            vol_d.copy_(vols_h[0])
            s16, p16 = pred.predict_device(vol_d, return_prob=True)
            s16, p16 = s16.clone(), p16.clone()
            s32, p32 = pred32.predict_device(vol_d, return_prob=True)
            inter = float(((s16 > 0) & (s32 > 0)).sum())
            res['parity_vs_fp32_path'] = dict(
                max_abs_softmax_diff=float((p16 - p32).abs().max()),
                voxels_changing_class=int((s16 != s32).sum()), voxels=int(s32.numel()),
                dice_vs_fp32_mask=2 * inter / (float((s16 > 0).sum()) + float((s32 > 0).sum()) + 1e-6))
            del eng32, pred32
        also_on = args.also == 'auto' and world == 1 and args.precision == 'f16x2' and not by_patch
        extra_nt = {}
        if also_on:
            # every other mode on the driver's record: short runs, same protocol, after the headline's timed region
            also = {}
            m, e32, vd32, vh32 = measure_inference_mode(plan, sd, device, 'fp32', 'fp32', 1, 3, 1, tta, args.shape, None)
            m['workload'] = 'configs[1] on the exact-fp32 kernels, mirroring %s' % ('on' if tta else 'off')
            m['dtype'] = 'f32'
            also['fp32_exact'] = m
            p32 = SlidingWindowPredictor(e32, do_mirroring=tta, batch=8)
            vd32.copy_(vh32[0])
            s32, q32 = p32.predict_device(vd32, return_prob=True)
            ref32 = (s32.clone(), q32.clone())
            vol_d.copy_(vols_h[0])
            sh, ph = pred.predict_device(vol_d, return_prob=True)
            inter = float(((sh > 0) & (ref32[0] > 0)).sum())
            res['parity_vs_fp32_path'] = dict(
                max_abs_softmax_diff=float((ph - ref32[1]).abs().max()), voxels_changing_class=int((sh != ref32[0]).sum()), voxels=int(sh.numel()),
                dice_vs_fp32_mask=2 * inter / (float((sh > 0).sum()) + float((ref32[0] > 0).sum()) + 1e-6))
            if args.cpu_baseline != 'off':
                s3, p3 = SlidingWindowPredictor(e32, do_mirroring=False, batch=8).predict_device(vd32, return_prob=True)
                extra_nt['fp32_exact'] = (s3.cpu().numpy(), p3.cpu().numpy())
            del e32, p32
            torch.cuda.empty_cache()
            m, e16, _, _ = measure_inference_mode(plan, sd, device, 'bf16', 'bf16', 8, 2, 1, tta, args.shape, ref32)
            m['workload'] = ('configs[2]: bf16 forward on channels-last bf16 tensors (wide levels), fused InstanceNorm+LeakyReLU, batch of 8 '
                             'synthetic volumes per step, mirroring %s' % ('on' if tta else 'off'))
            m['dtype'] = 'bf16 operands and activation tensors / f32 accumulate'
            also['configs2_bf16_storage'] = m
            del e16, ref32
            torch.cuda.empty_cache()
            for prec in ('fp32', 'f16x2'):
                also['train_' + prec] = measure_training_mode(plan, sd, device, prec, 8, 2)
                torch.cuda.empty_cache()
            if args.cpu_baseline != 'off':
                tp = training_parity_tiny()
                for prec in ('fp32', 'f16x2'):
                    also['train_' + prec]['parity'] = tp[prec]
            res['also'] = also
        if world == 1 and args.cpu_baseline != 'off':
            P = plan.patch_size
            vol_d.copy_(vols_h[0])
            # the oracle check runs on the engine of the mode when the mode claims fp32-grade results (fp32, f16x2)
            eng32 = eng if args.precision in ('fp32', 'f16x2') else UNetEngine(plan, sd, device)
            x = vol_d[None, :, :P[0], :P[1], :P[2]].contiguous()
            lg = eng32.forward_logits(x).cpu().numpy()
            p_nt = SlidingWindowPredictor(eng32, do_mirroring=False, batch=8)
            seg_nt, prob_nt = p_nt.predict_device(vol_d, return_prob=True)
            gpu_nt = (seg_nt.cpu().numpy(), prob_nt.cpu().numpy())
            res['cpu_baseline'], res['parity'], px = cpu_baseline(plan, sd, vols_h[0].numpy(), tta, args.cpu_baseline, gpu_nt, lg, extra_nt)
            for k, v in px.items():
                res['also'][k]['parity_vs_oracle'] = v
        if also_on and args.cli != 'off':
            # the boundary: the commands themselves, as children, after everything this process measures (its tensors are released first)
            del pred
            eng._bufs.clear()
            gc.collect()
            torch.cuda.empty_cache()
            nm_kernel = measure_no_mirroring_rate(plan, sd, device, args.shape)
            res['cli'] = measure_cli(plan, sd, args.shape)
            res['cli']['kernel_only'] = dict(mirroring_volumes_per_s=round(res['value'], 3), no_mirroring_volumes_per_s=nm_kernel,
                                             train_ms_f16x2=res['also']['train_f16x2']['ms_per_step'])
        if legs is None and also_on and args.legs == 'auto':
            del eng
            gc.collect()
            torch.cuda.empty_cache()
            legs = legs_child()
        if legs:
            res['legs'] = legs
        detail = args.detail_file
        if detail in ('none', 'off', ''):
            detail = None
        elif detail == 'auto':
            detail = os.path.join(ROOT, 'gpurun_out', 'bench_detail.json') if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else None
        if detail:
            try:
                with open(detail, 'w') as f:
                    json.dump(res, f, indent=1)
            except OSError:
                pass
        print(json.dumps(compact_line(res)), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(argv)
    return run_rank(args)


if __name__ == '__main__':
    sys.exit(main())
