"""Stand-in for paderbox.utils.random_utils (only imported, never sampled, by the golden script)."""


class _Sampler:
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __call__(self, *a, **k):
        raise NotImplementedError('augmentation samplers are out of scope for the goldens')


class TruncatedExponential(_Sampler):
    pass


class Uniform(_Sampler):
    pass


class LogUniform(_Sampler):
    pass


class LogTruncatedNormal(_Sampler):
    pass
