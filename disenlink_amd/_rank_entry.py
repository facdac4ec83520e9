"""Rank entry of ``python -m disenlink_amd.main --gpus N`` under torch.distributed.run.  The CLI's own arguments travel in
the environment (DL_MAIN_ARGV, JSON): torch.distributed.run's argument parser classifies every ``--option`` on its command
line before it reaches the script's remainder, and the reference's ``--run`` is an ambiguous prefix of its ``--run-path``.
Started by disenlink_amd/launch.py, one process per rank."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from disenlink_amd.main import main
    main(json.loads(os.environ["DL_MAIN_ARGV"]))
