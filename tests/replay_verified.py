#!/usr/bin/env python3
"""tools/synthetic_replay.py with every proof checked by the oracle's verifier restatement (test
infrastructure: the only place the replay touches oracle/). Same arguments as the tool:

    python tests/replay_verified.py --every 10
    python tests/replay_verified.py --every 1 --out gpurun_out/replay_full.json
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle_lib
    orc = oracle_lib.load()
    spec = importlib.util.spec_from_file_location("synthetic_replay", os.path.join(ROOT, "tools", "synthetic_replay.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    tool.main(sys.argv[1:], verify=orc.verify)


if __name__ == "__main__":
    main()
