"""vitamd — host side of the MI355X-native ViT training path (ctypes over libvitamd.so)."""
