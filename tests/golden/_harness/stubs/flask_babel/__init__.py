"""Harness stub for flask_babel (labels only; no arithmetic)."""


def lazy_gettext(s, **kw):
    return s % kw if kw else s


gettext = lazy_gettext


def get_locale():
    return None


class Babel:
    def __init__(self, *a, **k):
        pass
