"""The multi-GPU collective: one all-reduce of the parameter gradient per step, RCCL called
directly through the C ABI (aph_comm_* in csrc/comm.hip).  One process per GPU; the 128-byte RCCL unique id is the only
thing that travels out of band -- through an already-initialised torch.distributed group when there is one (bench.py under
torchrun: a one-off control-plane broadcast), through a rendezvous file keyed by the launch (clip_fft.py --ranks spawns the
ranks itself), or through a throw-away gloo group on torchrun's rendezvous (torchrun + clip_fft.py).
"""
import ctypes
import os
import time

import torch

from . import _ffi


class Comm:
    def __init__(self, rank, world, uid, lib=None):
        self.lib = lib if lib is not None else _ffi.lib()
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(uid))
        self.lib.call('aph_comm_init', self.rank, self.world, ctypes.cast(buf, ctypes.c_void_p), ctypes.byref(h))
        self.handle = h

    def all_reduce_(self, t, stream=None):
        """in-place sum over the ranks of a contiguous f32 CUDA tensor, asynchronous on `stream` (default: torch's current)"""
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError('all_reduce_ expects a contiguous f32 CUDA tensor')
        st = ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if stream is None else stream
        self.lib.call('aph_allreduce_f32', self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(), st)
        return t

    def ranks_seen(self):
        """ncclCommCount of the live communicator (bench.py's `rccl_ranks_seen`)"""
        n = ctypes.c_int(0)
        self.lib.call('aph_comm_ranks', self.handle, ctypes.byref(n))
        return n.value

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.cdll.aph_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def new_unique_id(lib=None):
    L = lib if lib is not None else _ffi.lib()
    buf = (ctypes.c_char * 128)()
    L.call('aph_comm_unique_id', ctypes.cast(buf, ctypes.c_void_p))
    return bytes(buf)


def _file_exchange(rank, world, uid, key, timeout=300.0):
    """rank 0 publishes the id in a rendezvous file, the others wait for it (all ranks share one node's /tmp).  `key` must be
    unique to this launch (clip_fft.py --ranks passes parent pid + a random port): no freshness heuristics, a slow importer
    simply finds the file already there."""
    # APH_RDV_DIR: a private (0700) directory created by the spawning parent and removed after the ranks exit (clip_fft.py / illustrip.py /
    # bench.py); without one (torchrun, where no common parent of ours exists) the launch-keyed name in TMPDIR
    path = os.path.join(os.environ.get('APH_RDV_DIR') or os.environ.get('TMPDIR', '/tmp'), 'aph_rccl_uid_%s' % key)
    if rank == 0:
        tmp = path + '.%d' % os.getpid()
        with open(tmp, 'wb') as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout:
        if os.path.isfile(path) and os.path.getsize(path) == 128:
            with open(path, 'rb') as f:
                return f.read()
        time.sleep(0.02)
    raise RuntimeError('no RCCL unique id at %s after %.0f s (is rank 0 running with the same rendezvous key?)' % (path, timeout))


def _gloo_exchange(rank, world, uid):
    """torchrun without a process group of the caller's: a throw-away gloo group on torchrun's own rendezvous (env://) carries
    the 128 bytes; nothing of it survives this call"""
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        t = torch.tensor(list(uid), dtype=torch.uint8) if rank == 0 else torch.zeros(128, dtype=torch.uint8)
        dist.broadcast(t, src=0)
        return bytes(t.tolist())
    finally:
        dist.destroy_process_group()


def spawn_ranks(fn, args, nranks):
    """Run `fn(local_rank, *args)` in `nranks` processes of this node (torch.multiprocessing.spawn) with a private 0700 rendezvous
    directory for the RCCL unique id, exported to the children as APH_RDV_DIR and removed when they have exited."""
    import shutil
    import tempfile
    import torch.multiprocessing as mp
    rdv = tempfile.mkdtemp(prefix='aph_rdv_')          # mode 0700: nobody else can plant or read the id
    os.environ['APH_RDV_DIR'] = rdv
    try:
        mp.spawn(fn, args=args, nprocs=nranks, join=True)
    finally:
        os.environ.pop('APH_RDV_DIR', None)
        shutil.rmtree(rdv, ignore_errors=True)


def create(rank, world, device=None, lib=None, key=None):
    """Comm over `world` ranks of this node.  Must be called by every rank with its GPU current."""
    if device is not None:
        torch.cuda.set_device(device)
    uid = new_unique_id(lib) if rank == 0 else None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() == world:
        dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
        t = torch.tensor(list(uid), dtype=torch.uint8, device=dev) if rank == 0 else torch.zeros(128, dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)                    # one-off control-plane exchange of 128 bytes
        uid = bytes(t.cpu().tolist())
    elif key is not None or 'APH_RUN_ID' in os.environ:
        uid = _file_exchange(rank, world, uid, key or '%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.environ['APH_RUN_ID']))
    elif dist.is_available() and all(k in os.environ for k in ('MASTER_ADDR', 'MASTER_PORT', 'RANK', 'WORLD_SIZE')):
        uid = _gloo_exchange(rank, world, uid)
    else:
        raise RuntimeError('comm.create: no way to hand the RCCL unique id to the other ranks (no torch.distributed group, no APH_RUN_ID '
                           'rendezvous key, no torchrun environment)')
    return Comm(rank, world, uid, lib)


# ---- supervised fallback ladder for multi-rank launches --------------------------------------------------------------------------------
# A rank that hangs inside a collective (or inside a graph capture that contains one) raises nothing: the only party that can end it is its
# PARENT.  So every rank process of a multi-rank launch is a supervisor that runs the real work in a child of its own, one "rung" (mode of
# operation) at a time, each rung under a wall budget; the supervisors of one launch agree through small files in a directory they share
# (one node) on whether a rung succeeded on EVERY rank -- all or none -- and otherwise move to the next rung together.  bench.py uses it
# so that `--gpus N` cannot end without its JSON line; it is independent of what the workers do.

def _kill_group(p):
    import signal
    try:
        os.killpg(p.pid, signal.SIGKILL)        # the worker was started in a session of its own: RCCL helper processes / threads go with it
    except (ProcessLookupError, PermissionError):
        pass
    try:
        p.wait(timeout=10)
    except Exception:
        pass


def _die_with_parent():
    """preexec of a worker: SIGKILL when its supervisor goes away (a launcher that kills the rank processes must not leave their workers behind)"""
    try:
        ctypes.CDLL(None).prctl(1, 9)           # PR_SET_PDEATHSIG, SIGKILL
    except Exception:
        pass


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def ladder(rank, world, rungs, make_cmd, sup_dir, budget_s=240.0, grace_s=30.0, log=None):
    """Run `rungs` (a list of names) in order until one succeeds on every rank.

    make_cmd(k, name, port) -> (argv, extra_env) of THIS rank's worker for rung k; its stdout goes to <sup_dir>/r<k>.rank<r>.out.
    Every rank's supervisor must call this with the same rungs / budget and a `sup_dir` they all see (created if missing).
    Returns (k or None, records, path of this rank's stdout file for rung k): records[k] = dict(rung, status per rank, seconds).
    A worker counts as failed when it exits non-zero or is still running after `budget_s` (it is then killed with its process group)."""
    import subprocess
    os.makedirs(sup_dir, exist_ok=True)
    log = log or (lambda s: None)
    records = []

    def wait_file(path, timeout):
        t0 = time.time()
        while time.time() - t0 < timeout:
            if os.path.isfile(path):
                with open(path) as f:
                    txt = f.read()
                if txt.endswith('\n'):
                    return txt.strip()
            time.sleep(0.05)
        return None

    def publish(path, txt):
        tmp = '%s.%d.tmp' % (path, os.getpid())
        with open(tmp, 'w') as f:
            f.write(txt + '\n')
        os.replace(tmp, path)

    for k, name in enumerate(rungs):
        t_rung = time.time()
        port_file = os.path.join(sup_dir, 'r%d.port' % k)
        if rank == 0:
            publish(port_file, str(free_port()))
        port = wait_file(port_file, budget_s + grace_s)
        if port is None:
            status = 'fail: no rendezvous port from rank 0'
        else:
            argv, extra = make_cmd(k, name, int(port))
            env = dict(os.environ)
            env.update(extra)
            out_path = os.path.join(sup_dir, 'r%d.rank%d.out' % (k, rank))
            with open(out_path, 'w') as out:
                p = subprocess.Popen(argv, env=env, stdout=out, start_new_session=True, preexec_fn=_die_with_parent)
                import signal
                prev = signal.signal(signal.SIGTERM, lambda *_: (_kill_group(p), os._exit(143)))     # a terminated supervisor takes its worker along
                try:
                    rc = p.wait(timeout=budget_s)
                    status = 'ok' if rc == 0 else 'fail: exit code %d' % rc
                except subprocess.TimeoutExpired:
                    _kill_group(p)
                    status = 'fail: still running after %.0f s, killed' % budget_s
                finally:
                    signal.signal(signal.SIGTERM, prev)
        publish(os.path.join(sup_dir, 'r%d.rank%d.status' % (k, rank)), status)
        # all or none: a rung counts only if EVERY rank's worker finished (a missing status = that supervisor is gone)
        deadline = t_rung + budget_s + grace_s
        sts = []
        for r in range(world):
            s = wait_file(os.path.join(sup_dir, 'r%d.rank%d.status' % (k, r)), max(deadline - time.time(), 1.0))
            sts.append(s if s is not None else 'fail: no status from its supervisor')
        records.append(dict(rung=name, status=sts, seconds=round(time.time() - t_rung, 1)))
        if all(s == 'ok' for s in sts):
            return k, records, os.path.join(sup_dir, 'r%d.rank%d.out' % (k, rank))
        log('rank %d: rung %d (%s) failed: %s' % (rank, k, name, '; '.join('rank %d %s' % (r, s) for r, s in enumerate(sts) if s != 'ok')))
    return None, records, None
