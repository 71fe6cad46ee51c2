"""Drop-in module paths of the reference package (``CGx.KNPEMI.*``) backed by cgx_hip."""
