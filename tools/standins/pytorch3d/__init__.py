"""Minimal stand-in for pytorch3d 0.7.4 (environment.yml:141) -- only what the
reference's hot path touches; semantics recalled from the published library."""
