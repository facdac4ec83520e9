"""Script form of ``python -m disenlink_amd.main`` for torch.distributed.run: its argument parser takes a SCRIPT PATH
followed by the script's arguments verbatim, whereas after ``-m module`` it still tries to match the module's own options
(``--run`` is "ambiguous" to it).  Started by disenlink_amd/launch.py, one process per rank."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    from disenlink_amd.main import main
    main(sys.argv[1:])
