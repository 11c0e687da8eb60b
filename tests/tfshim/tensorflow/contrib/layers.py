"""TEST INFRASTRUCTURE: tf.contrib.layers (imported by the reference, never called on this path)."""
