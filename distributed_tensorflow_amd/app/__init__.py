"""``app`` namespace: flag registry and ``app.run`` program entry.

The reference calls ``tf.app.run(main=main, argv=...)`` at ``tf_distributed.py:163`` with an
undefined ``main``; the evident intent -- parse flags, then call ``main(argv)`` and exit
with its return value -- is what ``run`` does.
"""
from __future__ import annotations

import sys
from typing import Callable, List, Optional

from . import flags  # noqa: F401
from .flags import FLAGS


def run(main: Optional[Callable[[List[str]], Optional[int]]] = None,
        argv: Optional[List[str]] = None) -> None:
    """Parse registered flags out of ``argv`` then call ``main(remaining_argv)``."""
    args = list(sys.argv if argv is None else argv)
    rest = FLAGS._parse(args[1:], handle_help=True)
    if main is None:
        main = getattr(sys.modules.get("__main__"), "main", None)
        if main is None:
            raise ValueError("app.run() needs a main function")
    sys.exit(main(args[:1] + rest))
