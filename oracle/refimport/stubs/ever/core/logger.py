import logging


def get_logger(name="ever-stub"):
    return logging.getLogger(name)
