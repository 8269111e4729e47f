"""CPU tests of bench.py's launch plumbing for N > 1: one process per GPU or a loud failure, the rank's host
thread bound next to its GPU's NUMA node, no fixed rendezvous port."""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_parse_cpulist():
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench.parse_cpulist("") == [] and bench.parse_cpulist("5") == [5]


def _fake_sysfs(tmp_path, bdf, node, cpulist):
    d = tmp_path / "bus" / "pci" / "devices" / bdf
    d.mkdir(parents=True)
    (d / "numa_node").write_text(f"{node}\n")
    if node >= 0:
        n = tmp_path / "devices" / "system" / "node" / f"node{node}"
        n.mkdir(parents=True)
        (n / "cpulist").write_text(cpulist + "\n")
    return str(tmp_path)


def test_cpus_near_pci_device_reads_sysfs(tmp_path):
    root = _fake_sysfs(tmp_path, "0000:c1:00.0", 1, "64-127,192-255")
    node, cpus = bench.cpus_near_pci_device("0000:C1:00.0", root)  # (HIP prints the id in either case)
    assert node == 1 and cpus[0] == 64 and cpus[-1] == 255 and len(cpus) == 128
    assert bench.cpus_near_pci_device("0000:05:00.0", root) == (-1, [])  # no such device: unbound, not an error


def test_no_numa_information_means_unbound(tmp_path):
    root = _fake_sysfs(tmp_path, "0000:03:00.0", -1, "")
    assert bench.cpus_near_pci_device("0000:03:00.0", root) == (-1, [])


class _Lib:
    def __init__(self, bdf):
        self.bdf = bdf

    def tg_device_pci_bus_id(self, dev, buf, n):
        ctypes.memmove(buf, self.bdf.encode() + b"\0", len(self.bdf) + 1)
        return 0


def test_bind_near_device_sets_the_affinity_of_this_process(tmp_path, monkeypatch):
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) < 2:
        pytest.skip("one allowed cpu")
    half = mine[: len(mine) // 2]
    root = _fake_sysfs(tmp_path, "0000:c1:00.0", 0, ",".join(str(c) for c in half))
    try:
        what = bench.bind_near_device(_Lib("0000:c1:00.0"), 0, root)
        assert sorted(os.sched_getaffinity(0)) == half and what.startswith("0000:c1:00.0 node 0 cpus") and f"({len(half)})" in what
        monkeypatch.setenv("TG_BENCH_BIND", "0")
        os.sched_setaffinity(0, mine)
        assert bench.bind_near_device(_Lib("0000:c1:00.0"), 0, root).startswith("off") and sorted(os.sched_getaffinity(0)) == mine
    finally:
        os.sched_setaffinity(0, mine)
    monkeypatch.delenv("TG_BENCH_BIND")
    assert bench.bind_near_device(object(), 0, root).startswith("unbound")  # (a library without the entry point)


def test_more_ranks_than_devices_is_an_error_unless_asked_for():
    assert bench.device_for_rank(3, 8, 8, False) == (3, None)
    dev, err = bench.device_for_rank(3, 8, 1, False)
    assert dev == -1 and "--share-devices" in err and "8" in err
    assert bench.device_for_rank(3, 8, 2, True) == (1, None)
    assert bench.device_for_rank(0, 1, 0, False)[1].startswith("no HIP device")


def test_world_without_master_port_is_refused(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(SystemExit, match="MASTER_PORT"):
        bench.Dist()
