"""CPU: the sharding / fan-out / reduction plumbing of physicl_amd.multidev.MultiDevice on a stand-in for the HIP
contexts (no GPU, no kernels): what goes to which context, and how the answers are put together again."""
import numpy as np

from physicl_amd.dist import shard_range
from physicl_amd.multidev import MultiDevice


class FakeDevice:
    made = []

    def __init__(self, device):
        self.device, self.ids, self.cols, self.closed, self.lib = device, np.zeros(0, np.int64), {}, False, None
        FakeDevice.made.append(self)

    count = property(lambda self: len(self.ids))
    capacity = property(lambda self: getattr(self, "_cap", 0))
    slots = property(lambda self: len(self.ids) + 3)
    np_dtype = np.float64

    def store_alloc(self, capacity, dtype="f64"):
        self._cap = capacity

    def fill_photons(self, n, id_base, c, e_min, e_max, seed):
        self.ids = np.arange(id_base, id_base + n, dtype=np.int64)
        self.cols = {0: self.ids * 10.0}

    def upload_state(self, state):
        n = len(state["E"])
        self.ids = np.arange(state["id_base"], state["id_base"] + n, dtype=np.int64)
        self.cols = {0: np.asarray(state["r"])[:, 0].astype(float), 12: np.asarray(state["E"], dtype=float)}

    def download(self, field, n=None, offset=0):
        n = self.count - offset if n is None else n
        return self.cols[field][offset:offset + n]

    def download_ids(self, n=None, offset=0):
        n = self.count - offset if n is None else n
        return self.ids[offset:offset + n]

    def download_state(self):
        return {"r": [self.cols[0], self.cols[0], self.cols[0]], "E": self.cols.get(12, self.cols[0]), "id": self.ids}

    def upload_rand(self, which, host):
        self.rand = (which, np.array(host))

    def step_fused(self, dt, scatter, planes, sync, lazy):
        return {"N": self.count, "sign": np.array([self.count, 0, 1]), "planes": np.zeros(0, np.int64), "hits": self.device + 1}

    def step_fused_multi(self, dt, k, scatter, planes, sync, raw):
        rows = np.tile(np.array([self.count, 1, 2, 3, self.device + 1], dtype=np.int64), (k, 1))
        return rows if raw else [{"N": int(r[0]), "sign": r[1:4].copy(), "planes": r[4:4].copy(), "hits": int(r[4])} for r in rows]

    def step_delete_flags(self, flags):
        keep = np.asarray(flags) == 0
        removed = int((~keep).sum())
        self.ids = self.ids[keep]
        self.cols = {k: v[keep] for k, v in self.cols.items()}
        return self.count, removed

    def is_uniform(self):
        return True

    def close(self):
        self.closed = True


class FakeHip:
    Device = FakeDevice


def test_shards_are_contiguous_index_blocks_with_global_ids():
    FakeDevice.made.clear()
    md = MultiDevice([0, 1, 2], hip=FakeHip)
    md.store_alloc(10)
    assert [d.capacity for d in FakeDevice.made] == [4, 4, 4] and md.capacity == 12      # ceil(10 / 3) slots on every shard
    md.fill_photons(10, 100, 1.0, 1.0, 1.0, 0)
    assert [list(d.ids) for d in FakeDevice.made] == [[100, 101, 102], [103, 104, 105], [106, 107, 108, 109]]
    assert [shard_range(10, g, 3) for g in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert md.count == 10 and md.slots == 19 and md.is_uniform()
    assert list(md.download_ids()) == list(range(100, 110))
    assert list(md.download_ids(4, 2)) == [102, 103, 104, 105]          # a window across two shards
    assert list(md.download(0, 3, 6)) == [1060.0, 1070.0, 1080.0] and len(md.download(0, 0, 10)) == 0
    # a window that starts in the last shard: the shards before it are asked for nothing AT OFFSET 0 (the window's
    # own offset can lie beyond an earlier shard's capacity, which the library's range check refuses even for n = 0)
    assert md._window(2, 7) == [(0, 0), (0, 0), (2, 1)] and list(md.download_ids(2, 7)) == [107, 108]
    md.close()
    assert all(d.closed for d in FakeDevice.made)


def test_a_store_takes_every_count_up_to_its_capacity_whatever_the_split():
    """the shard sizes of shard_range(n, g, G) are not monotone in n -- 8 objects on 5 devices are (1, 2, 1, 2, 2),
    7 are (1, 1, 2, 1, 2) -- so shards sized for 8 could not take a re-upload of 7.  Every shard gets ceil(capacity / G)."""
    FakeDevice.made.clear()
    md = MultiDevice([0] * 5, hip=FakeHip)
    md.store_alloc(8)
    caps = [d.capacity for d in FakeDevice.made]
    assert caps == [2] * 5 and md.capacity == 10
    for n in range(1, md.capacity + 1):
        sizes = [shard_range(n, g, 5)[1] - shard_range(n, g, 5)[0] for g in range(5)]
        assert sum(sizes) == n and all(sz <= c for sz, c in zip(sizes, caps)), (n, sizes)
    md.close()


def test_uploads_are_split_and_counters_are_summed():
    FakeDevice.made.clear()
    md = MultiDevice([0, 0], hip=FakeHip)
    n = 7
    md.store_alloc(n)
    md.upload_state({"r": np.arange(3 * n, dtype=float).reshape(n, 3), "E": np.arange(n) + 0.5, "id_base": 40})
    a, b = FakeDevice.made
    assert list(a.ids) == [40, 41, 42] and list(b.ids) == [43, 44, 45, 46] and list(b.cols[12]) == [3.5, 4.5, 5.5, 6.5]
    s = md.download_state()
    assert list(s["id"]) == list(range(40, 47)) and list(s["r"][0]) == [0.0, 3.0, 6.0, 9.0, 12.0, 15.0, 18.0]
    md.upload_rand(2, np.arange(n) / 10.0)
    assert list(a.rand[1]) == [0.0, 0.1, 0.2] and list(b.rand[1]) == [0.3, 0.4, 0.5, 0.6]
    o = md.step_fused(1e-3, None, (), True, True)
    assert o["N"] == 7 and list(o["sign"]) == [7, 0, 2] and o["hits"] == 2
    raw = md.step_fused_multi(1e-3, 3, {}, (), True, True)
    assert raw.shape == (3, 5) and list(raw[0]) == [7, 2, 4, 6, 2]
    rows = md.step_fused_multi(1e-3, 2, {}, (), True, False)
    assert rows[1]["N"] == 7 and list(rows[1]["sign"]) == [2, 4, 6] and rows[1]["hits"] == 2
    # flags are cut at the shards' CURRENT counts, and later windows follow the new counts
    assert md.step_delete_flags([1, 0, 0, 0, 1, 1, 0]) == (4, 3)
    assert list(md.download_ids()) == [41, 42, 43, 46] and md._counts() == [2, 2]
    md.close()
