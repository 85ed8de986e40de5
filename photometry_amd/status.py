# -*- coding: utf-8 -*-
"""
``STATUS`` enum with the reference's integer values (photometry/BasePhotometry.py:48-59);
these integers are what the C-ABI per-target ``status`` arrays carry and what
``todolist.status`` stores (photometry/taskmanager.py:538-541).
"""
import enum


@enum.unique
class STATUS(enum.Enum):
	UNKNOWN = 0  #: The status is unknown. The actual calculation has not started yet.
	STARTED = 6  #: The calculation has started, but not yet finished.
	OK = 1       #: Everything has gone well.
	ERROR = 2    #: Encountered a catastrophic error that I could not recover from.
	WARNING = 3  #: Something is a bit fishy.
	ABORT = 4    #: The calculation was aborted.
	SKIPPED = 5  #: The target was skipped because the algorithm found that to be the best solution.
