"""Harness stub: in-process dict cache."""


class SimpleCache:
    def __init__(self, *a, **k):
        self._d = {}

    def get(self, k):
        return self._d.get(k)

    def set(self, k, v, timeout=None):
        self._d[k] = v
        return True
