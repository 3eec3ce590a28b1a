"""`python -m harry_amd.cli IN OUT [...]` -- a thin alias of the `harry` executable (harry_amd/bin/harry, built from
harry_amd/csrc/cli/main.cpp over the C ABI): the reference's command line `harry IN OUT [-l L] [-a A] [-q Q] [-c] [-f hry|ply]
[--ply-ascii]` (main.cc:23-123) plus the additive --profile / --chunk / --device / --shards."""
from __future__ import annotations

import os
import subprocess
import sys

HARRY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "harry")


def main(argv=None) -> int:
    if not os.path.exists(HARRY):
        raise FileNotFoundError(f"{HARRY} is missing: build it with `make -C harry_amd/csrc`")
    return subprocess.call([HARRY] + list(sys.argv[1:] if argv is None else argv))


if __name__ == "__main__":
    sys.exit(main())
