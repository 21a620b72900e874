"""Interval / end triggers (``padertorch/train/trigger.py:8-142``)."""
import copy

__all__ = ['IntervalTrigger', 'EndTrigger']


class IntervalTrigger:
    """True once per ``period`` iterations / epochs (including index 0).

    >>> t = IntervalTrigger(2, 'iteration')
    >>> [t(i, i // 3) for i in range(6)]
    [True, False, True, False, True, False]
    """

    @classmethod
    def new(cls, trigger):
        if isinstance(trigger, IntervalTrigger):
            return copy.deepcopy(trigger)
        assert len(trigger) == 2, trigger
        return cls(*trigger)

    def __init__(self, period, unit):
        assert isinstance(period, int), (type(period), period)
        assert unit in ('epoch', 'iteration'), unit
        self.period, self.unit = period, unit
        self.last = (-1, -1)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.period}, {self.unit})'

    def _index(self, iteration, epoch):
        return (epoch, self.last[1]) if self.unit == 'epoch' else (iteration, self.last[0])

    def __call__(self, iteration, epoch):
        index, last = self._index(iteration, epoch)
        if last == index:
            return False
        self.set_last(iteration, epoch)
        return (index % self.period) == 0

    def set_last(self, iteration, epoch):
        self.last = (iteration, epoch)


class EndTrigger(IntervalTrigger):
    """True from ``period`` on.

    >>> t = EndTrigger(2, 'epoch')
    >>> [t(i, i // 3) for i in range(8)]
    [False, False, False, False, False, False, True, True]
    """

    def __call__(self, iteration, epoch):
        index = epoch if self.unit == 'epoch' else iteration
        return index >= self.period
