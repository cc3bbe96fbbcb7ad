#!/usr/bin/env python3
"""The reference's literal entry point: `python3 basecall.py fast5_dir fasta_dir [flags]`, run from a directory that holds
`models/` (radian/basecall.py:28-30,143-144; README.md:56-59) -- same flags, defaults, relative artefact paths and output.
Everything lives in radian_amd.basecall (the reference's loop body on the MI355X); this file only makes the command a RADIAN
user types work unchanged, from any working directory (the package is found beside this file, not through the cwd)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from radian_amd.basecall import main  # noqa: E402

if __name__ == "__main__":
    main()
