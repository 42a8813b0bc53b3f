"""Global command-line flag registry: ``DEFINE_string`` / ``DEFINE_integer`` / ... / ``FLAGS``.

The reference selects a process's role with two flags declared at import time and read
lazily (``tf_distributed.py:14-18``): the first attribute access on ``FLAGS`` parses
``sys.argv``. Both ``--flag=value`` and ``--flag value`` are accepted, unknown arguments are
left for the program (``app.run`` forwards them to ``main``), and booleans also accept the
``--noflag`` spelling. ``--help`` / ``-h`` / ``--helpfull`` on the command line print the flags
declared so far and exit when the flags are first read -- before a ps task would start serving.
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, List, Optional


class _FlagDef:
    __slots__ = ("name", "default", "help", "parser", "is_bool")

    def __init__(self, name: str, default: Any, help: str, parser: Callable[[str], Any],
                 is_bool: bool = False):
        self.name = name
        self.default = default
        self.help = help
        self.parser = parser
        self.is_bool = is_bool


def _parse_bool(text: str) -> bool:
    low = text.strip().lower()
    if low in ("1", "true", "t", "yes", "y"):
        return True
    if low in ("0", "false", "f", "no", "n"):
        return False
    raise ValueError("not a boolean: %r" % (text,))


class _FlagValues:
    """Lazily parsed flag namespace."""

    def __init__(self):
        self.__dict__["_defs"] = {}        # name -> _FlagDef
        self.__dict__["_values"] = {}      # name -> parsed/overridden value
        self.__dict__["_parsed"] = False
        self.__dict__["_unparsed"] = []

    # -- registration --------------------------------------------------------------------
    def _define(self, flag: _FlagDef) -> None:
        self._defs[flag.name] = flag
        # a flag defined after the first parse still picks its value from argv
        if self._parsed:
            self.__dict__["_parsed"] = False

    # -- parsing -------------------------------------------------------------------------
    def _parse(self, argv: Optional[List[str]] = None,
               handle_help: Optional[bool] = None) -> List[str]:
        if handle_help is None:
            handle_help = argv is None          # the process's own command line
        args = list(sys.argv[1:] if argv is None else argv)
        defs: Dict[str, _FlagDef] = self._defs
        values: Dict[str, Any] = {}
        rest: List[str] = []
        i = 0
        while i < len(args):
            arg = args[i]
            i += 1
            if arg == "--":
                rest.extend(args[i:])
                break
            if not arg.startswith("-") or arg in ("-",):
                rest.append(arg)
                continue
            body = arg.lstrip("-")
            if handle_help and body in ("help", "h", "helpfull", "helpshort") \
                    and body not in defs:
                print("flags of %s:\n%s" % (sys.argv[0], self.get_help()))
                sys.exit(0)
            name, has_eq, text = body.partition("=")
            name = name.replace("-", "_") if name.replace("-", "_") in defs else name
            if name in defs:
                flag = defs[name]
                if has_eq:
                    values[name] = flag.parser(text)
                elif flag.is_bool:
                    # "--flag" alone is true; "--flag false" also accepted
                    if i < len(args) and args[i].lower() in ("true", "false", "0", "1"):
                        values[name] = _parse_bool(args[i])
                        i += 1
                    else:
                        values[name] = True
                else:
                    if i >= len(args):
                        raise ValueError("flag --%s needs a value" % name)
                    values[name] = flag.parser(args[i])
                    i += 1
            elif name.startswith("no") and name[2:] in defs and defs[name[2:]].is_bool \
                    and not has_eq:
                values[name[2:]] = False
            else:
                rest.append(arg)
        # a parse starts from the defaults; only values assigned explicitly survive it
        for name in [n for n in self._values if not self.__dict__.get("_frozen_" + n)]:
            del self._values[name]
        for name, val in values.items():
            if not self.__dict__.get("_frozen_" + name):  # explicit assignment wins
                self._values[name] = val
        self.__dict__["_parsed"] = True
        self.__dict__["_unparsed"] = rest
        return rest

    def _reset(self) -> None:
        """Forget every definition and value (test helper)."""
        self._defs.clear()
        self._values.clear()
        for key in [k for k in self.__dict__ if k.startswith("_frozen_")]:
            del self.__dict__[key]
        self.__dict__["_parsed"] = False
        self.__dict__["_unparsed"] = []

    # -- attribute protocol --------------------------------------------------------------
    def __getattr__(self, name: str) -> Any:
        if name.startswith("__"):
            raise AttributeError(name)
        if not self.__dict__["_parsed"]:
            self._parse()
        if name in self._values:
            return self._values[name]
        if name in self._defs:
            return self._defs[name].default
        raise AttributeError("unknown flag %r" % (name,))

    def __setattr__(self, name: str, value: Any) -> None:
        self._values[name] = value
        self.__dict__["_frozen_" + name] = True

    def __contains__(self, name: str) -> bool:
        return name in self._defs

    def flag_values_dict(self) -> Dict[str, Any]:
        if not self.__dict__["_parsed"]:
            self._parse()
        return {n: self._values.get(n, d.default) for n, d in self._defs.items()}

    def get_help(self) -> str:
        lines = []
        for name in sorted(self._defs):
            d = self._defs[name]
            lines.append("  --%s: %s (default: %r)" % (name, d.help, d.default))
        return "\n".join(lines)


FLAGS = _FlagValues()


def DEFINE_string(name: str, default: Optional[str], help: str = "") -> None:  # noqa: N802
    FLAGS._define(_FlagDef(name, default, help, str))


def DEFINE_integer(name: str, default: Optional[int], help: str = "") -> None:  # noqa: N802
    FLAGS._define(_FlagDef(name, default, help, int))


def DEFINE_float(name: str, default: Optional[float], help: str = "") -> None:  # noqa: N802
    FLAGS._define(_FlagDef(name, default, help, float))


def DEFINE_boolean(name: str, default: Optional[bool], help: str = "") -> None:  # noqa: N802
    FLAGS._define(_FlagDef(name, default, help, _parse_bool, is_bool=True))


DEFINE_bool = DEFINE_boolean


def define_cluster_flags(default_job: str = "worker") -> None:
    """Declare the role + cluster flags every example uses.

    ``--job_name`` / ``--task_index`` are the reference's flags (``tf_distributed.py:14-15``);
    ``--ps_hosts`` / ``--worker_hosts`` (with the reference's ``--ps`` / ``--worker``
    spellings, ``tf_distributed.py:137-148``) make the host lists configurable, which the
    reference declared but never used.
    """
    DEFINE_string("job_name", default_job, "Either 'ps' or 'worker'")
    DEFINE_integer("task_index", 0, "Index of task within the job")
    DEFINE_string("ps_hosts", "", "Comma-separated list of hostname:port pairs")
    DEFINE_string("worker_hosts", "", "Comma-separated list of hostname:port pairs")
    DEFINE_string("ps", "", "Alias of --ps_hosts")
    DEFINE_string("worker", "", "Alias of --worker_hosts")
