"""``time_limit`` (biolith/utils/misc.py:7-21): SIGALRM-based timeout used by ``fit(timeout=...)``."""
import signal
from contextlib import contextmanager


class TimeoutException(Exception):
    pass


@contextmanager
def time_limit(seconds):
    """Raise :class:`TimeoutException` in the main thread after ``seconds``.

    The HIP driver polls the device from Python while sampling, so the handler fires between polls
    and the driver then aborts the persistent kernel through its host-mapped flag.
    """

    def _handler(signum, frame):
        raise TimeoutException("Timed out")

    previous = signal.signal(signal.SIGALRM, _handler)
    signal.alarm(int(seconds))
    try:
        yield
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, previous)
