"""CPU, world_size 2 over gloo: clip sharding and the final gather of the N > 1 path."""
import json
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_and_gather_world2(tmp_path):
    out = str(tmp_path / "result.json")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "gloo_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    r = json.load(open(out))
    assert len(r["res"]) == 5
    for i, wav in enumerate(r["res"]):
        n = 100 + 37 * i - (i % 2)          # odd clips were converted by rank 1, which trims one sample
        assert wav == [3 * i] * n
    assert r["single"] == [list(range(10)), list(range(11))]
    # feature dump: clip i was written by rank i mod 2, every file exactly once, [T_h, 768] each
    assert r["owners"] == [0, 1, 0, 1, 0]
    assert r["shapes"] == [[10 * (i + 1), 768] for i in range(5)]
