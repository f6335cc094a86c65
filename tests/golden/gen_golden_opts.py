#!/usr/bin/env python
"""Pin `parse_options` for the five shipped option files (options/bench_*.toml = BASELINE configs[0..4]).

Runs the REFERENCE parser (/root/reference/neosr/utils/options.py:39-275) on each file, in the build
container only, and writes its output dict as tests/golden/opt_<name>.json (paths normalised, like
gen_golden.py:dump_opt).  tests/test_host_logic.py holds our own parser to these dumps.

    python tests/golden/gen_golden_opts.py
"""

from __future__ import annotations

import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))

from gen_golden import REF, dump_opt, install_reference  # noqa: E402


def main() -> None:
    files = sorted((ROOT / "options").glob("bench_*.toml"))
    install_reference(str(files[0]))
    from neosr.utils.options import parse_options

    for f in files:
        sys.argv = ["gen_golden_opts", "-opt", str(f)]
        opt, _ = parse_options(str(REF), is_train=True)
        dump_opt(opt, f"opt_{f.stem}.json")


if __name__ == "__main__":
    main()
