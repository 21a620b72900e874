import pathlib


class SummaryWriter:
    def __init__(self, logdir):
        self.logdir = pathlib.Path(logdir)
        self.logdir.mkdir(parents=True, exist_ok=True)
        (self.logdir / 'events.out.tfevents.0.shim').write_text('')

    def _noop(self, *args, **kwargs):
        pass

    add_scalar = add_histogram = add_audio = add_image = add_text = add_figure = _noop

    def close(self):
        pass
