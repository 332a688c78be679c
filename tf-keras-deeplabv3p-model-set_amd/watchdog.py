"""First-steps timeout for data-parallel runs.

The reference's MirroredStrategy (train.py:143-158) either trains or raises; a multi-process RCCL job can instead sit
forever in its first collective (a rank that never arrives, collectives captured into a hipGraph that the runtime replays
out of order, two communicators contending).  `FirstStepsGuard` bounds the part of a world > 1 run where that can show
-- tracing the executor (every collective runs once, eagerly), the graph capture, the first replays -- and on expiry
prints what to try and ends the process with a non-zero code.  It never re-executes anything: a process that has
touched the GPU must not exec (the caller's launcher starts a fresh job with the suggested switch).

DL3P_DIST_TIMEOUT_S: seconds (default 300; 0 disables)."""
import os
import sys
import threading

EXIT_CODE = 3

HINT = ('dl3p: the first data-parallel steps did not finish within %.0f s (rank %d of %d, stage: %s).\n'
        'The collectives are captured into the step\'s hipGraphs and SyncBatchNorm runs on its own communicator.\n'
        'Start the job again with one of:\n'
        '  DL3P_COLLECTIVES_IN_GRAPH=0   collectives issued eagerly between graph segments\n'
        '  DL3P_ONE_COMM=1               SyncBatchNorm statistics on the gradient communicator\n'
        '  DL3P_SYNC_BN=0                per-replica BatchNorm statistics (no forward collectives)\n'
        'Exiting with code %d; nothing is retried in this process.\n')


class FirstStepsGuard:
    """with FirstStepsGuard(rank, world, 'trace'): ...  -- arms a timer thread for the block; `stage(name)` renames the
    stage the message reports.  A world of 1 (or timeout 0) arms nothing."""

    def __init__(self, rank=0, world=1, stage='first step', timeout=None, _exit=os._exit, _out=None):
        t = os.environ.get('DL3P_DIST_TIMEOUT_S')
        self.timeout = float(t) if (timeout is None and t) else (300.0 if timeout is None else float(timeout))
        self.rank, self.world, self.name = rank, world, stage
        self._exit, self._out = _exit, _out
        self._timer = None

    def stage(self, name):
        self.name = name

    def _expired(self):
        out = self._out or sys.stderr
        out.write(HINT % (self.timeout, self.rank, self.world, self.name, EXIT_CODE))
        out.flush()
        self._exit(EXIT_CODE)

    def __enter__(self):
        if self.world > 1 and self.timeout > 0:
            self._timer = threading.Timer(self.timeout, self._expired)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None
        return False
