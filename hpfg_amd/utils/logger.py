"""Logging helper with the reference's call shape (utils/logger.py:15-34): stdout + a log file."""
import logging
import os


def _get_logger(filename, level="info"):
    log = logging.getLogger(filename)
    if log.handlers:
        return log
    log.setLevel({"debug": logging.DEBUG, "info": logging.INFO, "warning": logging.WARNING, "error": logging.ERROR}.get(level, logging.INFO))
    fmt = logging.Formatter("%(asctime)s - %(levelname)s: %(message)s")
    sh = logging.StreamHandler()
    sh.setFormatter(fmt)
    log.addHandler(sh)
    try:
        os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
        fh = logging.FileHandler(filename, encoding="utf-8")
        fh.setFormatter(fmt)
        log.addHandler(fh)
    except OSError:
        pass
    return log
