"""Host-side logic that needs no GPU."""
import pytest



def test_lazy_results_behave_like_a_dict():
    """NeRFSystem.forward returns the blended colours as entries that are computed on first access (models/nerf_system.py:
    136-142 have one reader, the validation PSNR): every way of reading the dict sees them, nothing runs before that."""
    from upnerf_amd.nerf_system import _LazyResults
    calls = []
    r = _LazyResults({"a": 1})
    r.lazy("b", lambda: calls.append("b") or 2)
    r.lazy("c", lambda: calls.append("c") or 3)
    assert "b" in r and "c" in r and "z" not in r and len(r) == 3 and not calls
    assert r.get("z", 7) == 7 and not calls
    assert r["b"] == 2 and calls == ["b"]
    assert r["b"] == 2 and calls == ["b"]          # once
    assert dict(r.items()) == {"a": 1, "b": 2, "c": 3} and calls == ["b", "c"]
    r3 = _LazyResults({"b": 1})
    r3.lazy("b", lambda: 2)                         # (r4 ADVICE) a pending thunk for a key that already holds a value counts once
    assert len(r3) == 1 and r3["b"] == 2 and len(r3) == 1
    r2 = _LazyResults({"a": 1})
    r2.lazy("b", lambda: 5)
    r2["b"] = 9                                     # an explicit value replaces the thunk
    assert r2["b"] == 9 and sorted(r2.keys()) == ["a", "b"] and list(r2.values()).count(9) == 1


def test_lazy_results_follow_the_whole_dict_protocol():
    """r3 ADVICE: pop / copy / update / setdefault / == / pickling must see the lazily evaluated entries too."""
    import pickle
    from upnerf_amd.nerf_system import _LazyResults

    def make():
        calls = []
        r = _LazyResults({"a": 1})
        r.lazy("b", lambda: calls.append("b") or 2)
        return r, calls

    r, calls = make()
    assert len(r) == 2 and "b" in r and calls == []
    assert r.pop("b") == 2 and calls == ["b"] and "b" not in r
    r, _ = make()
    assert r.copy() == {"a": 1, "b": 2} and type(r.copy()) is dict
    r, _ = make()
    assert r == {"a": 1, "b": 2}
    r, _ = make()
    assert r.setdefault("b", 7) == 2
    r, _ = make()
    assert sorted(r.keys()) == ["a", "b"] and sorted(r.values()) == [1, 2] and dict(r.items()) == {"a": 1, "b": 2}
    r, _ = make()
    back = pickle.loads(pickle.dumps(r))
    assert back == {"a": 1, "b": 2} and type(back) is dict
    r, _ = make()
    r.update({"b": 5})
    assert r["b"] == 5


def test_zero_pool_serves_one_arena_per_step_and_falls_back_off_plan():
    """zero_pool: the second step's requests are slices of one zeroed arena (256-byte aligned, disjoint); a request that
    leaves the previous step's sequence, and any request outside a step, is a plain torch.zeros."""
    import torch
    from upnerf_amd import zero_pool
    zero_pool._S.plan = None
    dev = torch.device("cpu")
    assert zero_pool.zeros(5, dev).shape == (5,)  # outside a step
    with zero_pool.step(dev):
        a, b = zero_pool.zeros(10, dev), zero_pool.zeros(100, dev)
        assert a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()  # nothing planned yet: two fills
    with zero_pool.step(dev):
        a, b = zero_pool.zeros(10, dev), zero_pool.zeros(100, dev)
        assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
        assert b.data_ptr() - a.data_ptr() == 256 and a.numel() == 10 and b.numel() == 100
        a += 1.0
        assert float(b.abs().max()) == 0.0
        c = zero_pool.zeros(7, dev)  # one more than planned
        assert c.untyped_storage().data_ptr() != a.untyped_storage().data_ptr() and float(c.abs().max()) == 0.0
    with zero_pool.step(dev):
        a = zero_pool.zeros(10, dev)
        x = zero_pool.zeros(33, dev)  # off the plan (100 was next): fills from here on
        y = zero_pool.zeros(7, dev)
        assert x.untyped_storage().data_ptr() != a.untyped_storage().data_ptr()
        assert y.untyped_storage().data_ptr() not in (a.untyped_storage().data_ptr(), x.untyped_storage().data_ptr())
    with zero_pool.step(dev):  # ... and the new sequence is the plan now
        a, x, y = zero_pool.zeros(10, dev), zero_pool.zeros(33, dev), zero_pool.zeros(7, dev)
        assert a.untyped_storage().data_ptr() == x.untyped_storage().data_ptr() == y.untyped_storage().data_ptr()
        assert float(torch.cat([a, x, y]).abs().max()) == 0.0
    try:
        with zero_pool.step(dev):
            zero_pool.zeros(3, dev)
            raise KeyError("a step that raised")
    except KeyError:
        pass
    assert zero_pool._S.rec is None and zero_pool._S.plan[1] == (10, 33, 7)  # the failed step changed nothing
