"""Harness stub for python-dotenv."""


def load_dotenv(*a, **k):
    return None
