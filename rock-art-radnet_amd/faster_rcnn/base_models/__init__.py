"""Backbone modules selected by Config.network (train.py:145-151, RADNet.py:727-733)."""
