"""Several GPUs in ONE process: ``Simulation(devices=[0, 1, ...])``.

The reference is a single process -- one ``threading.Thread`` running the loop, one lock, scripts and notebooks that
poll ``get_state()`` (physicl/__init__.py:400-432, 501-541) -- so "drops into the existing examples" needs a way to
use a node's GPUs without ``torchrun``.  ``MultiDevice`` is that way: it looks like one ``_hip.Device`` to the
simulation and owns one library context (one HIP device, one stream, one particle store) per entry of ``devices``:

* particles are sharded by global index in contiguous blocks (``dist.shard_range``: device g of G owns
  ``[g*N/G, (g+1)*N/G)``); ids are global, and the device RNG is keyed by the id, so every row and every photon's
  history is exactly what one device would have produced;
* every call fans out to the contexts on a small thread pool (ctypes releases the GIL: the launches run side by side)
  and the returned counters -- alive, hits / removed, sign counts, plane crossings -- are summed on the host.  That sum
  IS the collective of this mode: the counter blocks are already in host memory when a launch returns, so no RCCL is
  involved (one process per GPU + RCCL all-reduce remains the other way to run, physicl_amd/dist.py);
* downloads concatenate in shard order (shards are contiguous index blocks and compaction is stable, so that is
  particle order), uploads are split the same way, and ``len(sim.objects)`` / ``get_state()`` are global.

Host-drawn randoms (``rng="numpy"``) work here -- unlike across processes -- because the one host stream is drawn
once and split by the shards' current counts: a seeded run still reproduces the reference's stream position.

Testable on a one-GPU box: ``devices=[0, 0]`` makes two contexts on the same GPU.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .dist import shard_range


class MultiDevice:
    """The subset of ``_hip.Device`` the host layer uses, over ``len(devices)`` contexts."""

    def __init__(self, devices, hip=None):
        from . import _hip
        self._hip = hip or _hip
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices must name at least one HIP device")
        self.shards = [self._hip.Device(d) for d in self.devices]
        self._pool = ThreadPoolExecutor(max_workers=len(self.shards), thread_name_prefix="pcl-dev")
        self._bounds = [(0, 0)] * len(self.shards)       # global index block of each shard at the last upload / fill
        self.lib = self.shards[0].lib
        self.device = self.devices[0]

    # ---------------------------------------------------------------- plumbing
    def _each(self, fn, *per_shard):
        """fn(shard, *args_for_that_shard) on every context, concurrently; results in shard order."""
        if len(self.shards) == 1:
            return [fn(self.shards[0], *[a[0] for a in per_shard])]
        futs = [self._pool.submit(fn, s, *[a[g] for a in per_shard]) for g, s in enumerate(self.shards)]
        return [f.result() for f in futs]

    def _counts(self):
        return [s.count for s in self.shards]

    def _split(self, host, counts=None):
        """A dense per-particle array cut at the shards' current counts."""
        counts = self._counts() if counts is None else counts
        host = np.asarray(host)
        if host.shape[0] != sum(counts):
            raise ValueError("one entry per particle: got %d for %d" % (host.shape[0], sum(counts)))
        out, at = [], 0
        for c in counts:
            out.append(host[at:at + c])
            at += c
        return out

    @staticmethod
    def _sum_dicts(outs):
        if outs[0] is None:
            return None
        tot = {}
        for k, v in outs[0].items():
            if isinstance(v, str):
                tot[k] = v
            elif isinstance(v, np.ndarray):
                tot[k] = np.sum([o[k] for o in outs], axis=0)
            else:
                tot[k] = sum(o[k] for o in outs)
        return tot

    # ---------------------------------------------------------------- lifecycle
    def close(self):
        for s in self.shards:
            s.close()
        self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._each(lambda s: s.sync())

    def info(self):
        infos = [s.info() for s in self.shards]
        return dict(infos[0], devices=infos, name=" + ".join(i["name"] for i in infos))

    def set_rtc_background(self, on=True):
        for s in self.shards:
            s.set_rtc_background(on)

    def rtc_wait(self):
        return sum(self._each(lambda s: s.rtc_wait()))

    # Level-1 helpers of CLProgram work on gathered host data, not on the store: the first context serves them
    def array(self, host, dtype=np.float64):
        return self.shards[0].array(host, dtype)

    def empty(self, n, dtype=np.float64):
        return self.shards[0].empty(n, dtype)

    def user_kernel(self, name, params, body):
        return self.shards[0].user_kernel(name, params, body)

    # ---------------------------------------------------------------- store
    def store_alloc(self, capacity, dtype="f64"):
        # every shard gets ceil(capacity / G): the shard sizes of shard_range(n, g, G) are not monotone in n (8 objects on 5
        # devices are (1, 2, 1, 2, 2), 7 are (1, 1, 2, 1, 2)), so a re-upload of fewer objects into the same store must
        # find room for the largest shard any n <= capacity can produce
        G = len(self.shards)
        self._each(lambda s: s.store_alloc(max(-(-capacity // G), 1), dtype))

    def store_free(self):
        self._each(lambda s: s.store_free())

    def reserve_compaction(self):
        self._each(lambda s: s.reserve_compaction())

    @property
    def np_dtype(self):
        return self.shards[0].np_dtype

    @property
    def capacity(self):
        """Objects the store takes whatever their number's split over the shards is: G x the smallest shard."""
        caps = [s.capacity for s in self.shards]
        return len(caps) * min(caps)

    @property
    def count(self):
        return sum(self._counts())

    @property
    def slots(self):
        return sum(s.slots for s in self.shards)

    def is_uniform(self):
        return all(s.is_uniform() for s in self.shards)

    def fill_photons(self, n, id_base, c, e_min, e_max, seed):
        G = len(self.shards)
        self._bounds = [shard_range(n, g, G) for g in range(G)]
        self._each(lambda s, b: s.fill_photons(b[1] - b[0], id_base + b[0], c, e_min, e_max, seed), self._bounds)

    def fill_photons_table(self, n, id_base, c, cdf, grid, seed):
        G = len(self.shards)
        self._bounds = [shard_range(n, g, G) for g in range(G)]
        self._each(lambda s, b: s.fill_photons_table(b[1] - b[0], id_base + b[0], c, cdf, grid, seed), self._bounds)

    def upload_state(self, state):
        n, G = len(np.asarray(state["E"])), len(self.shards)
        self._bounds = [shard_range(n, g, G) for g in range(G)]
        base = int(state.get("id_base", 0))

        def part(b):
            lo, hi = b
            sub = {"id_base": base + lo}
            for k in ("r", "v", "dr", "dv"):             # (n, 3), as _hip.Device.upload_state takes them
                if state.get(k) is not None:
                    sub[k] = np.asarray(state[k]).reshape(n, 3)[lo:hi]
            for k in ("E", "id", "kind"):
                if state.get(k) is not None:
                    sub[k] = np.asarray(state[k])[lo:hi]
            return sub
        self._each(lambda s, b: s.upload_state(part(b)), self._bounds)

    def upload(self, field, host, offset=0):
        """Elements [offset, offset + len(host)) of one field, in global particle order, to the shards that hold them."""
        host = np.ascontiguousarray(host)
        at = 0
        for s, (n, off) in zip(self.shards, self._window(len(host), offset)):
            if n:
                s.upload(field, host[at:at + n], off)
                at += n

    def upload_rand(self, which, host):
        self._each(lambda s, h: s.upload_rand(which, np.ascontiguousarray(h)), self._split(host))

    RAND3_CHUNK = 1 << 20

    def upload_rand3(self, u3, offset=0):
        """A chunk of the global (n, 3) draw: its rows go to the shards that own particles [offset, offset + len(u3))."""
        u3 = np.ascontiguousarray(u3, dtype=np.float64).reshape(-1, 3)
        at = 0
        for s, c in zip(self.shards, self._counts()):
            lo, hi = max(offset, at), min(offset + len(u3), at + c)
            if hi > lo:
                s.upload_rand3(u3[lo - offset:hi - offset], lo - at)
            at += c

    def upload_kind(self, host, offset=0):
        if offset:
            raise NotImplementedError("partial kind uploads on a multi-device store")
        self._each(lambda s, h: s.upload_kind(np.ascontiguousarray(h)), self._split(host))

    def _concat(self, parts, dtype=None):
        return np.concatenate(parts) if len(parts) > 1 else parts[0]

    def _window(self, n, offset):
        """(per-shard n, per-shard offset) of the global window [offset, offset + n)."""
        counts = self._counts()
        total = sum(counts)
        n = total - offset if n is None else n
        out, at = [], 0
        for c in counts:
            lo, hi = max(offset, at), min(offset + n, at + c)
            # (a shard wholly outside the window: n = 0 at offset 0 -- the window's own offset may lie beyond its capacity)
            out.append((hi - lo, lo - at) if hi > lo else (0, 0))
            at += c
        return out

    def download(self, field, n=None, offset=0):
        return self._concat(self._each(lambda s, w: s.download(field, w[0], w[1]), self._window(n, offset)))

    def download_ids(self, n=None, offset=0):
        return self._concat(self._each(lambda s, w: s.download_ids(w[0], w[1]), self._window(n, offset)))

    def download_kind(self, n=None, offset=0):
        return self._concat(self._each(lambda s, w: s.download_kind(w[0], w[1]), self._window(n, offset)))

    def download_state(self):
        parts = self._each(lambda s: s.download_state())
        out = {}
        for k in parts[0]:
            if isinstance(parts[0][k], list):
                out[k] = [self._concat([p[k][j] for p in parts]) for j in range(len(parts[0][k]))]
            else:
                out[k] = self._concat([p[k] for p in parts])
        return out

    # ---------------------------------------------------------------- steps: fan out, sum the counters
    def step_newton(self, dt):
        self._each(lambda s: s.step_newton(dt))

    def step_scatter_isotropic(self, *a, **kw):
        hits = self._each(lambda s: s.step_scatter_isotropic(*a, **kw))
        return None if hits[0] is None else sum(hits)

    def step_scatter_delete(self, *a, **kw):
        outs = self._each(lambda s: s.step_scatter_delete(*a, **kw))
        return sum(o[0] for o in outs), sum(o[1] for o in outs)

    def scatter_pcoll(self, *a, **kw):
        return self._concat(self._each(lambda s: s.scatter_pcoll(*a, **kw)))

    def step_delete_flags(self, flags):
        outs = self._each(lambda s, f: s.step_delete_flags(np.ascontiguousarray(f)), self._split(np.asarray(flags)))
        return sum(o[0] for o in outs), sum(o[1] for o in outs)

    def step_counters(self, planes=()):
        return np.sum(self._each(lambda s: s.step_counters(planes)), axis=0)

    def plane_energies(self, plane, n_hint=None):
        return self._concat(self._each(lambda s: s.plane_energies(plane)))      # (a shard does not know its share of the hint)

    def step_fused(self, dt, scatter=None, planes=None, sync=True, lazy=False):
        return self._sum_dicts(self._each(lambda s: s.step_fused(dt, scatter, planes, sync, lazy)))

    def step_fused_read(self, n_planes=0):
        return self._sum_dicts(self._each(lambda s: s.step_fused_read(n_planes)))

    def last_scatter_hits(self):
        return sum(self._each(lambda s: s.last_scatter_hits()))

    def step_fused_delete(self, *a, **kw):
        return self._sum_dicts(self._each(lambda s: s.step_fused_delete(*a, **kw)))

    def _sum_rows(self, outs, raw):
        if raw:
            return np.sum(outs, axis=0)
        return [self._sum_dicts([o[k] for o in outs]) for k in range(len(outs[0]))]

    def step_fused_multi(self, dt, k_steps, scatter, planes=(), sync=True, raw=False):
        outs = self._each(lambda s: s.step_fused_multi(dt, k_steps, scatter, planes, sync, raw))
        return None if outs[0] is None else self._sum_rows(outs, raw)

    def trace_ahead(self, ids, *a, defer=False, **kw):
        """Every context is asked for every tracked id and answers NaN rows for the particles it does not hold; a particle
        lives in exactly one shard, so the first answer that is not NaN is the row."""
        outs = self._each(lambda s: s.trace_ahead(ids, *a, defer=defer, **kw))

        def merge(parts):
            rows = parts[0]
            for o in parts[1:]:
                rows = np.where(np.isnan(rows[:, :, :1]), o, rows)
            return rows
        if not defer:
            return merge(outs)
        return lambda: merge([read() for read in outs])

    def step_fused_delete_multi(self, dt, k_steps, A, n, seed=0, step=0, planes=None, raw=False):
        return self._sum_rows(self._each(lambda s: s.step_fused_delete_multi(dt, k_steps, A, n, seed, step, planes, raw)), raw)

    def step_mixed_multi(self, dt, k_passes, phases, scatter=None, delete=None, planes=(), seed=0, step=0, raw=False):
        return self._sum_rows(self._each(lambda s: s.step_mixed_multi(dt, k_passes, phases, scatter, delete, planes, seed, step, raw)), raw)
