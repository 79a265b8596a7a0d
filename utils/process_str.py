"""Text post-processing of evaluation.py --post_processing=True (reference utils/process_str.py:6-47): keep ASCII
letters and spaces, lower-case; each helper takes a string or a list of strings."""
import re


def list_operation(text, func):
    if isinstance(text, str):
        return func(text)
    if isinstance(text, list):
        return [func(t) for t in text]
    raise Exception(f"unsupported type {type(text)}")


def filter_ascii_str(text):
    return re.sub(r"[^a-zA-Z ]", "", text)


def filter_ascii_text(text):
    return list_operation(text, filter_ascii_str)


def convert_lower_text(text):
    return list_operation(text, lambda t: t.lower())
