class _Bar:
    def progress(self, v):
        pass


def progress(v):
    return _Bar()
