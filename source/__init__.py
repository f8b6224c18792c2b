"""Import-path compatibility with the reference: its YAML configs name classes by `source.*` paths
(configs/poco.yaml:28,41, configs/ppsurf.yaml:5,13).  Every module here re-exports the MI355X-native implementation."""
