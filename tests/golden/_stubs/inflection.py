"""Stand-in for inflection.underscore (CamelCase -> snake_case)."""
import re


def underscore(word):
    word = re.sub(r'([A-Z]+)([A-Z][a-z])', r'\1_\2', word)
    word = re.sub(r'([a-z\d])([A-Z])', r'\1_\2', word)
    return word.replace('-', '_').lower()
