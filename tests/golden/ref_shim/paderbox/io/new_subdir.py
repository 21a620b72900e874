def get_new_subdir(*args, **kwargs):
    raise NotImplementedError('not on the hot path')
