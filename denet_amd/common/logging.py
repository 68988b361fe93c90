"""print-like logging with the interface of denet/common/logging.py (:1-88): debug / verbose (level 15) / info / warning /
error / critical take any number of arguments; `add_arguments(parser)` adds `--log-level`, `init(args)` configures the
"denet" logger and logs the command line."""
import logging as _logging
import sys

VERBOSE = 15
_logger = None
_flush = False


def _msg(*args):
    return " ".join(str(a) for a in args)


def _get():
    if _logger is None:
        init()
    return _logger


def _emit(level, args):
    _get().log(level, _msg(*args))
    if _flush:
        sys.stdout.flush()


def verbose_enabled():
    """True once a driver has configured the logger (init) and it passes VERBOSE records: per-step timing lines of the hot
    path (denet_sparse.py:145) are formatted only then"""
    return _logger is not None and _logger.isEnabledFor(VERBOSE)


def debug(*args):
    _emit(_logging.DEBUG, args)


def verbose(*args):
    _emit(VERBOSE, args)


def info(*args):
    _emit(_logging.INFO, args)


def warning(*args):
    _emit(_logging.WARNING, args)


def error(*args):
    _emit(_logging.ERROR, args)


def critical(*args):
    _emit(_logging.CRITICAL, args)


def exception(*args):
    _get().exception(_msg(*args))


def setLevel(lvl):
    if str(lvl).upper() == "VERBOSE":
        lvl = VERBOSE
    _get().setLevel(lvl)


def add_arguments(parser):
    parser.add_argument("--log-level", default="verbose", help="Log level")


def init(args=None, flush=False):
    global _logger, _flush
    _logging.basicConfig(stream=sys.stdout, format="%(message)s")
    _logging.addLevelName(VERBOSE, "VERBOSE")
    _logger = _logging.getLogger("denet")
    level = "VERBOSE" if args is None else str(getattr(args, "log_level", "verbose")).upper()
    _logger.setLevel(VERBOSE if level == "VERBOSE" else level)
    _flush = bool(flush)
    info("--------------------------------")
    info("Program Cmdline: " + " ".join(sys.argv))
    info("--------------------------------")
