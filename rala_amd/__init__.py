"""rala_amd — MI355X-native pile-o-gram + transitive-reduction hot path of rvaser/rala."""
