"""ORACLE -- CPU restatement of the reference's algorithms for the Faster R-CNN hot path.

Test infrastructure only: nothing under rock-art-radnet_amd/ (the product) imports it.
  glue.py   NumPy restatement of rpn.py / utils.py / RADNet.py host glue (pinned by goldens)
  dense.py  NumPy restatement of the Keras/TF graph: conv, frozen BN, pooling, RoI resize,
            dense heads, losses, Adam, with explicit backward ("parity unpinned": TF absent)
"""
