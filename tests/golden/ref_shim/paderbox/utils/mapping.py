class DispatchError(KeyError):
    pass


class Dispatcher(dict):
    def __getitem__(self, item):
        try:
            return super().__getitem__(item)
        except KeyError:
            raise DispatchError(f'Invalid option {item!r}.') from None
