"""profiles/pmc_traffic.json from the rocprofv3 --pmc passes over tools/pmc_traffic.py:
   python tools/make_pmc_traffic.py <root> <label>
expects <root>/pmc_<mode>_FETCH_SIZE and <root>/pmc_<mode>_WRITE_SIZE for mode in fp32, bf16, bf16_storage, f16x2 (missing modes are
skipped).  Per mode: the dominant convolution kernel's mean FETCH_SIZE / WRITE_SIZE per launch (KiB), FETCH_SIZE corrected by the
calibration launch of the same run (a streaming kernel over 1 GiB in + 1 GiB out: MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys

KERNEL = {'fp32': 'conv3d_k3_ws_kernel<ConvCfg<1, 32, 1, 4, 8', 'bf16': 'conv3d_k3_bf16_ws_kernel<ConvCfg<1, 32',
          'bf16_storage': 'cl_conv3_kernel', 'f16x2': '1, 8, 1, 4, 6, 3, 1>, false, false, true>'}      # f16x2: the X3 instance of the 32-row configuration (the bench's timed launches)
CAL = 'instnorm_lrelu_apply_kernel'
CAL_KIB = 1024 * 1024


def read(d):
    rows = []
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def mean_of(rows, needle, counter, grid_min=0):
    v = [float(r['Counter_Value']) for r in rows if needle in r['Kernel_Name'] and r['Counter_Name'] == counter]
    return (sum(v) / len(v), len(v), v) if v else (None, 0, [])


root, label = sys.argv[1], sys.argv[2]
out = {}
for mode, needle in KERNEL.items():
    fd, wd = os.path.join(root, 'pmc_%s_FETCH_SIZE' % mode), os.path.join(root, 'pmc_%s_WRITE_SIZE' % mode)
    if not (os.path.isdir(fd) and os.path.isdir(wd)):
        continue
    fr, wr = read(fd), read(wd)
    names = sorted({r['Kernel_Name'] for r in fr if needle in r['Kernel_Name']})
    cal_f, _, _ = mean_of(fr, CAL, 'FETCH_SIZE')
    cal_w, _, _ = mean_of(wr, CAL, 'WRITE_SIZE')
    f, nf, fv = mean_of(fr, needle, 'FETCH_SIZE')
    w, nw, wv = mean_of(wr, needle, 'WRITE_SIZE')
    if f is None or w is None or not cal_f:
        print('mode %s: kernel or calibration launch not found (%s)' % (mode, needle))
        continue
    fcorr, wcorr = CAL_KIB / cal_f, CAL_KIB / cal_w
    # forwards in the run: 2 (tools/pmc_traffic.py); per-launch mean over all launches of the kernel
    out[mode] = dict(kernel=names[0][:160] if names else needle, launches=nf, fetch_kib_mean=round(f, 1), write_kib_mean=round(w, 1),
                     fetch_correction=round(fcorr, 4), write_correction=round(wcorr, 4),
                     calibration='%s over 1 GiB in + 1 GiB out read FETCH_SIZE %.0f KiB, WRITE_SIZE %.0f KiB' % (CAL, cal_f, cal_w),
                     traffic_bytes_per_launch=int(round((f * fcorr + w * wcorr) * 1024)),
                     fetch_kib_per_launch=[round(x) for x in fv[:len(fv) // 2 or len(fv)]],
                     write_kib_per_launch=[round(x) for x in wv[:len(wv) // 2 or len(wv)]],
                     measured=label)
    print(mode, json.dumps({k: out[mode][k] for k in ('launches', 'fetch_kib_mean', 'write_kib_mean', 'fetch_correction',
                                                      'write_correction', 'traffic_bytes_per_launch')}))
out['how'] = ('rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_traffic.py (WMH_PMC_MODE = '
              'the key; tools/profile_round.sh), forwards of the BASELINE network at WMH_BATCH = 16 patches -- what the bench\'s predictor puts into one '
              'forward; mean over every launch of the named kernel; corrections from the calibration launch of the same run')
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_traffic.json')
json.dump(out, open(dst, 'w'), indent=2)
print('wrote', dst)
