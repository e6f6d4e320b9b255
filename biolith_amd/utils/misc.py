"""``time_limit`` -- the alarm behind ``fit(timeout=...)`` (reference counterpart: biolith/utils/misc.py:7-21).

The reference bounds ``mcmc.run`` with SIGALRM.  A signal cannot stop a running GPU kernel, so here the
alarm is only the messenger: ``fit`` polls the device from Python while the persistent kernel samples, the
interval timer's handler raises between two polls, and ``fit`` then stops the kernel through its
host-mapped abort flag (``bl_nuts_abort``) before re-raising.
"""
import signal
import threading


class TimeoutException(Exception):
    """Raised inside a ``time_limit`` block when its time is up (same name as the reference's exception)."""


class time_limit:
    """Context manager: raise :class:`TimeoutException` in the main thread once ``seconds`` have elapsed.

    Uses the real-time interval timer (fractional seconds are honoured) and restores the previous SIGALRM
    disposition on exit.  Outside the main thread signals cannot be delivered; the block then runs unarmed
    and ``fit`` relies on its own deadline while polling the device.
    """

    def __init__(self, seconds):
        self.seconds = float(seconds)
        self._armed = False
        self._previous = None

    def _expired(self, signum, frame):
        raise TimeoutException("Timed out")

    def __enter__(self):
        if threading.current_thread() is threading.main_thread() and self.seconds > 0:
            self._previous = signal.signal(signal.SIGALRM, self._expired)
            signal.setitimer(signal.ITIMER_REAL, self.seconds)
            self._armed = True
        return self

    def __exit__(self, exc_type, exc, tb):
        if self._armed:
            signal.setitimer(signal.ITIMER_REAL, 0.0)
            signal.signal(signal.SIGALRM, self._previous)
            self._armed = False
        return False
