"""`batch.shared_workspace`: one workspace per (device, stream), at most `MAX_SHARED_WORKSPACES` of them alive
(the cache was keyed by raw stream handles and never evicted - tens of GB per entry).  Host logic only:
a stand-in for the `torch.cuda` stream query, no buffer is allocated."""
import types

from python_stable_3d_truss_analysis_amd import batch


def _fake_torch(handle_box):
    stream = lambda device: types.SimpleNamespace(cuda_stream=handle_box[0])
    return types.SimpleNamespace(cuda=types.SimpleNamespace(current_stream=stream))


def test_shared_workspace_cache_is_bounded_and_lru(monkeypatch):
    monkeypatch.setattr(batch, "_SHARED_WORKSPACES", {})
    box = [0]
    torch = _fake_torch(box)
    made = {}
    for handle in range(1, batch.MAX_SHARED_WORKSPACES + 1):
        box[0] = handle
        made[handle] = batch.shared_workspace(torch, "cuda:0")
    box[0] = 2
    assert batch.shared_workspace(torch, "cuda:0") is made[2]          # same stream, same workspace
    assert batch.shared_workspace(torch, "cuda:1") is not made[2]      # another device: another one - and the
    assert len(batch._SHARED_WORKSPACES) == batch.MAX_SHARED_WORKSPACES   # least recently used entry (stream 1) went
    box[0] = 1
    assert batch.shared_workspace(torch, "cuda:0") is not made[1]
    box[0] = 2
    assert batch.shared_workspace(torch, "cuda:0") is made[2]          # recently used entries survive the evictions
    assert len(batch._SHARED_WORKSPACES) == batch.MAX_SHARED_WORKSPACES
