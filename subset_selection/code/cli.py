"""Drop-in for the reference's subset_selection/code/cli.py: same command line, MI355X hot path."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from acav100m_amd.subset_selection.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
