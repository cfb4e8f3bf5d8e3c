"""Harness stub: two-name provider, only used by the reference's debug printing."""


class Provider:
    first_names = {'Aino': 1, 'Eino': 1}
    last_names = {'Virtanen': 1, 'Korhonen': 1}
