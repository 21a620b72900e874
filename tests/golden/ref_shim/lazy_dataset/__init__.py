def from_list(items):
    return list(items)
