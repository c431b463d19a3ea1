"""Layer plugins of the MI355X backend (reference EM/plugins/).  Plugins are discovered by module name exactly like
the reference does; user plugins written against NumPy arrays work unchanged (layers are handed over as host arrays)."""
