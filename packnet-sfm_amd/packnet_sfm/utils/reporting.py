"""Small bookkeeping shared by models and losses: a dict of logs and a dict of detached scalar tensors."""


class Reporting:
    """Mixin: `self._report(kind)` returns the named dict, created on first use (kinds: logs / losses / metrics)."""

    def _report(self, kind):
        store = self.__dict__.setdefault('_reports', {})
        return store.setdefault(kind, {})

    def _record(self, kind, key, value):
        self._report(kind)[key] = value.detach()
