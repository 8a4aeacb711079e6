"""BaseSolver — mirror of the reference's src/solver.py:13-222 (`Solver(config, paras, mode)`,
`load_data / set_model / exec`, `backward`, `load_ckpt`, `save_checkpoint`, `verbose`, `progress`,
`write_log`; checkpoint layout {'model','optimizer','global_step',<metric>} unchanged so
checkpoints move between the two code bases).

MI355X differences:
  * there is no CPU path: the solver refuses `--cpu` and fails loudly when no gfx950 device or the
    HIP library is missing (the reference silently falls back to CPU, src/solver.py:29-30);
  * one process per GPU: when launched under torch.distributed.run (WORLD_SIZE > 1) `backward`
    routes through parallel.DataParallelEngine (bucketed RCCL all-reduce overlapped with BPTT);
    clipping and the NaN-skip then see the averaged gradients on every rank;
  * TensorBoard is optional (not in this image): scalars/text fall back to <logdir>/log.jsonl;
  * apex AMP (`--amp`) is rejected: the hot path is exact f32 by contract (BASELINE north_star).
"""
import abc
import json
import math
import os
import sys
import time

import torch
import yaml

from .. import _lib
from .option import default_hparas
from .util import human_format, Timer


class _JsonlWriter:
    ''' SummaryWriter stand-in: the three calls the solvers make, appended to log.jsonl '''

    def __init__(self, logdir, flush_secs=180):
        os.makedirs(logdir, exist_ok=True)
        self._f = open(os.path.join(logdir, 'log.jsonl'), 'a')
        self._flush_secs, self._last = flush_secs, time.time()

    def _put(self, rec):
        self._f.write(json.dumps(rec) + '\n')
        if time.time() - self._last > self._flush_secs:
            self._f.flush()
            self._last = time.time()

    def add_scalars(self, name, d, step):
        self._put({'step': step, 'name': name, 'scalars': {k: float(v) for k, v in d.items()}})

    def add_text(self, name, text, step):
        self._put({'step': step, 'name': name, 'text': text})

    def add_image(self, *args, **kwargs):
        pass

    def close(self):
        self._f.close()


def _make_writer(logdir, flush_secs):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(logdir, flush_secs=flush_secs)
    except ImportError:
        return _JsonlWriter(logdir, flush_secs)


class BaseSolver():
    ''' Prototype Solver: config - yaml-styled dict, paras - argparse outcome '''

    def __init__(self, config, paras, mode):
        self.config = config
        self.paras = paras
        self.mode = mode
        for k, v in default_hparas.items():
            setattr(self, k, v)
        if not self.paras.gpu:
            raise RuntimeError('--cpu is not supported: every operator of this path is a gfx950 kernel')
        if getattr(paras, 'amp', False):
            raise RuntimeError('--amp is not supported: the path computes in exact f32')
        if not torch.cuda.is_available():
            raise RuntimeError('no HIP device visible')
        _lib.load()                                   # fail here, loudly, if libasrk.so is missing
        # one process per GPU (torch.distributed.run sets these)
        self.rank = int(os.environ.get('RANK', 0))
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.local_rank = int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device('cuda', self.local_rank)
        self.dist, self.dp = None, None
        # ASRK_FORCE_DIST=1: take the data-parallel path (RCCL communicator, loss weights from the count all-reduce,
        # gradient buckets, collectives launched from the backward hooks) with ONE rank - how the multi-GPU code of
        # the product solver is exercised on a 1-GPU box (tests/test_parallel_gpu.py)
        self.force_dist = mode == 'train' and self.world == 1 and os.environ.get('ASRK_FORCE_DIST', '0') == '1'
        if (self.world > 1 or self.force_dist) and mode == 'train':
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', '29534')
                dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.device)
            self.dist = dist
        elif self.world > 1 and mode == 'test':
            # decoding shards utterances over the ranks; the only exchange is gathering the result rows
            # (python objects) on rank 0 -> a host-side gloo group is all it needs
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                dist.init_process_group('gloo')
            self.dist = dist
        self.amp = False

        self.exp_name = paras.name
        if self.exp_name is None:
            self.exp_name = paras.config.split('/')[-1].replace('.yaml', '')
            if mode == 'train':
                self.exp_name += '_sd{}'.format(paras.seed)
        self.emb_decoder = None     # embedding-fusion plugin: out of scope

        if mode == 'train':
            os.makedirs(paras.ckpdir, exist_ok=True)
            self.ckpdir = os.path.join(paras.ckpdir, self.exp_name)
            os.makedirs(self.ckpdir, exist_ok=True)
            self.logdir = os.path.join(paras.logdir, self.exp_name)
            self.log = _make_writer(self.logdir, self.TB_FLUSH_FREQ) if self.rank == 0 else None
            self.timer = Timer()
            self.step = 0
            self.valid_step = config['hparas']['valid_step']
            self.max_step = config['hparas']['max_step']
            self.verbose('Exp. name : {}'.format(self.exp_name))
            self.verbose('Loading data... large corpus may took a while.')
        elif mode == 'test':
            os.makedirs(paras.outdir, exist_ok=True)
            self.ckpdir = os.path.join(paras.outdir, self.exp_name)
            # the training config fixes the acoustic features, text encoder and model
            self.src_config = yaml.load(open(config['src']['config'], 'r'), Loader=yaml.FullLoader)
            self.paras.load = config['src']['ckpt']
            self.verbose('Evaluating result of tr. config @ {}'.format(config['src']['config']))

    def enable_data_parallel(self):
        ''' call at the end of set_model(): wraps self.model's gradients when WORLD_SIZE > 1 '''
        if self.dist is not None:
            from ..parallel import DataParallelEngine
            self.dp = DataParallelEngine(self.model, self.dist, force_collectives=self.force_dist)
            self.verbose('Data parallel | {} ranks over RCCL, {} gradient buckets'.format(
                self.world, len(self.dp._buckets)))

    ERR_POLL_STEPS = 50     # how often the training loops read device-side flags back (each read synchronises)

    def poll_device_errors(self, force=False):
        ''' Hand-off timeouts of the persistent kernels are sticky flags in their workspace and the NaN guard of the
            fused update is a device-side predicate; reading either synchronises the stream, so the training loops
            look every ERR_POLL_STEPS steps (plus the first step, before every validation pass, before every
            checkpoint and at the end of training), not every step. '''
        if not (force or self.step == 1 or self.step % self.ERR_POLL_STEPS == 0):
            return
        from .. import ops
        ops.check_errors()
        watch, self._nan_watch = getattr(self, '_nan_watch', []), []
        if watch:
            bad = torch.isnan(torch.stack([n.reshape(()) for _, n in watch])).tolist()
            for (step, _), b in zip(watch, bad):
                if b:
                    self.verbose('Error : grad norm is NaN @ step ' + str(step))

    def backward(self, loss):
        ''' backward + clip + (NaN-guarded) optimizer step (reference: src/solver.py:75-91).  With a fused optimiser
            the returned norm is a 0-d DEVICE tensor: nothing is read back here. '''
        self.timer.set()
        if self.dp is not None:
            self.dp.backward(loss)
        else:
            loss.backward()
        if getattr(self.optimizer, 'fused', False):
            # clipping is folded into the fused update: the gradients are read once, never rewritten
            from ..fused_optim import grad_norm_and_coef
            grad_norm, coef = grad_norm_and_coef(list(self.model.parameters()), self.GRAD_CLIP)
            if getattr(self.optimizer, 'device_nan_skip', False):
                # the update kernel itself leaves parameters and state untouched when the norm is NaN
                self.optimizer.step(grad_norm, self.GRAD_CLIP, coef=coef)
                if not hasattr(self, '_nan_watch'):
                    self._nan_watch = []
                self._nan_watch.append((self.step, grad_norm))
            elif math.isnan(grad_norm):
                self.verbose('Error : grad norm is NaN @ step ' + str(self.step))
            else:
                self.optimizer.step(grad_norm, self.GRAD_CLIP, coef=coef)
        else:
            grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.GRAD_CLIP)
            if math.isnan(grad_norm):
                self.verbose('Error : grad norm is NaN @ step ' + str(self.step))
            else:
                self.optimizer.step()
        self.timer.cnt('bw')
        return grad_norm

    def load_ckpt(self):
        ''' Load ckpt if --load option is specified (reference: src/solver.py:93-125) '''
        if self.paras.load:
            ckpt = torch.load(self.paras.load, map_location=self.device if self.mode == 'train' else 'cpu')
            self.model.load_state_dict(ckpt['model'])
            metric, score = "None", 0.0
            for k, v in ckpt.items():
                if type(v) is float:
                    metric, score = k, v
            if self.mode == 'train':
                self.step = ckpt['global_step']
                self.optimizer.load_opt_state_dict(ckpt['optimizer'])
                self.verbose('Load ckpt from {}, restarting at step {} (recorded {} = {:.2f} %)'.format(
                    self.paras.load, self.step, metric, score))
            else:
                self.model.eval()
                self.verbose('Evaluation target = {} (recorded {} = {:.2f} %)'.format(
                    self.paras.load, metric, score))

    def verbose(self, msg):
        if self.paras.verbose and self.rank == 0:
            for m in (msg if type(msg) == list else [msg]):
                print('[INFO]', m.ljust(100))

    def progress(self, msg):
        if self.paras.verbose and self.rank == 0:
            sys.stdout.write("\033[K")  # Clear line
            print('[{}] {}'.format(human_format(self.step), msg), end='\r')

    def write_log(self, log_name, log_dict):
        ''' scalars (dict), text (str) or images (tuple) to the logger (reference: src/solver.py:141-161) '''
        if self.log is None:
            return
        if type(log_dict) is dict:
            vals = {key: float(val.detach()) if torch.is_tensor(val) else val
                    for key, val in log_dict.items() if val is not None}
            log_dict = {key: val for key, val in vals.items() if not math.isnan(val)}
        if log_dict is None:
            pass
        elif len(log_dict) > 0:
            if 'align' in log_name or 'spec' in log_name:
                img, form = log_dict
                self.log.add_image(log_name, img, global_step=self.step, dataformats=form)
            elif 'text' in log_name or 'hyp' in log_name:
                self.log.add_text(log_name, log_dict, self.step)
            else:
                self.log.add_scalars(log_name, log_dict, self.step)

    def save_checkpoint(self, f_name, metric, score, show_msg=True):
        ''' (reference: src/solver.py:163-186) — rank 0 only.  Never from unverified state: the sticky hand-off
            flags are read first (every rank; it raises), so parameters updated from an aborted persistent kernel's
            gradients - or a validation pass that itself timed out - cannot reach latest.pth / best_*.pth. '''
        if self.mode == 'train':
            self.poll_device_errors(force=True)
        if self.rank != 0:
            return
        ckpt_path = os.path.join(self.ckpdir, f_name)
        full_dict = {
            "model": self.model.state_dict(),
            "optimizer": self.optimizer.get_opt_state_dict(),
            "global_step": self.step,
            metric: score
        }
        torch.save(full_dict, ckpt_path)
        if show_msg:
            self.verbose("Saved checkpoint (step = {}, {} = {:.2f}) and status @ {}".
                         format(human_format(self.step), metric, score, ckpt_path))

    # ----------------------------------- Abstract methods ------------------------------------------ #
    @abc.abstractmethod
    def load_data(self):
        raise NotImplementedError

    @abc.abstractmethod
    def set_model(self):
        raise NotImplementedError

    @abc.abstractmethod
    def exec(self):
        raise NotImplementedError
