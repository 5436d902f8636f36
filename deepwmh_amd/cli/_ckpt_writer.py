"""Helper process of `nnUNet_train`: turns parameter snapshots into nnU-Net checkpoint files while the trainer keeps the GPU busy.

An epoch of the reference's schedule (`-b 150`, /root/reference/deepwmh/pipeline/DCNN_multistage.py:292-309: 4.2 s here) ends with up to three
files of 125-250 MB (model_best, model_ep_NNNN with --save_every_epoch, model_latest with the momentum).  `torch.save` holds the interpreter
lock for most of their 0.2-0.4 s each, so written from a thread of the training process they starve the thread that launches kernels (measured:
44 instead of 30 ms per iteration in the epoch that follows) -- hence a process.  It never touches a GPU (it is started with the devices
hidden).  Protocol, one JSON object per line on stdin:
  {"op": "init", "order": [[name, shape], ...], "offsets": [...], "plans": <pickle file>}      parameter layout of the flat arena; the plans dict
  {"op": "save", "params": <raw float32 file>, "momentum": <file or null>, "folder", "fold", "name", "epoch", "extra": {...}}
  {"op": "drop", "files": [...]}                                                                snapshot files no later job needs
Every line is answered on stdout by "ok <name>" or "error <text>"; EOF ends the process after the queued lines.
"""
import json
import os
import pickle
import sys
from collections import OrderedDict


def main():
    import numpy as np
    from deepwmh_amd import checkpoint as ck
    order, offsets, plans = None, None, None
    out = sys.stdout
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            job = json.loads(line)
            if job['op'] == 'init':
                order, offsets = [(n, tuple(s)) for n, s in job['order']], [int(v) for v in job['offsets']]
                with open(job['plans'], 'rb') as f:
                    plans = pickle.load(f)
                out.write('ok init\n')
            elif job['op'] == 'save':
                flat = np.fromfile(job['params'], dtype=np.float32)
                sd = OrderedDict()
                for (name, shape), o in zip(order, offsets):
                    n = int(np.prod(shape))
                    sd[name] = flat[o:o + n].reshape(shape).copy()
                extra = dict(job.get('extra') or {})
                if job.get('momentum'):
                    extra['deepwmh_amd_momentum'] = np.fromfile(job['momentum'], dtype=np.float32)
                ck.save_checkpoint(job['folder'], sd, plans, job['fold'], job['name'], epoch=job['epoch'], extra=extra)
                out.write('ok %s\n' % job['name'])
            elif job['op'] == 'drop':
                for f in job['files']:
                    try:
                        os.remove(f)
                    except OSError:
                        pass
                out.write('ok drop\n')
            else:
                out.write('error unknown op %r\n' % job.get('op'))
        except Exception as e:                                    # reported to the trainer, which fails the command
            out.write('error %s: %s\n' % (type(e).__name__, str(e).replace('\n', ' ')))
        out.flush()
    return 0


if __name__ == '__main__':
    sys.exit(main())
